"""Evaluation side of the detection path: result formats, bbox COCOeval."""
from .coco import COCO
from .cocoeval import COCOeval, Params
from .results import (VISDRONE_CLASSES, ResultMerger, coco_records, detection_line, parse_detection_results,
                      parse_per_class, write_detection_results)

__all__ = ["COCO", "COCOeval", "Params", "VISDRONE_CLASSES", "ResultMerger", "coco_records", "detection_line",
           "parse_detection_results", "parse_per_class", "write_detection_results"]
