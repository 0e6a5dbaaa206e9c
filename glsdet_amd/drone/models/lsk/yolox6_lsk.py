"""Twin of drone/models/lsk/yolox6_lsk.py (text-identical to lsk/yolox6.py in the reference)."""
from glsdet_amd.drone.body import LskCrossYoloBody as YoloBody  # noqa: F401
