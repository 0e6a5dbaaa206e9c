"""Twin of drone/models/block/non_local/yolo_patch_nonlocal_plus.py: `YoloBody(num_classes,
phi)` (YOLOX with the Global-Local fusion neck), HIP backed."""
from glsdet_amd.drone.body import GLYoloBody as YoloBody  # noqa: F401
