"""`from mmdet.core import UnifiedForegroundPacking` (ufpmp_det_eval.py:11; ufp/mmdet/core/ufp/)."""
from glsdet_amd.ufp import unified_foreground_packing as UnifiedForegroundPacking  # noqa: F401
