"""`from pycocotools.cocoeval import COCOeval` (ufpmp_det_eval.py:13)."""
from glsdet_amd.eval import COCOeval  # noqa: F401
