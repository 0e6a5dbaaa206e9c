"""`from pycocotools.coco import COCO` (ufpmp_det_eval.py:12)."""
from glsdet_amd.eval import COCO  # noqa: F401
