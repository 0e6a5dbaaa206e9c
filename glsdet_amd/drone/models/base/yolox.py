"""Twin of drone/models/base/yolox.py: `YoloBody(num_classes, phi)` (YOLOX), HIP backed."""
from glsdet_amd.drone.body import BaseYoloBody as YoloBody  # noqa: F401
