"""Twin of drone/models/lsk/yolox6.py: `YoloBody(num_classes, phi)` = cross-scale decoupled head on the LSK-attention
CSPDarknet (lsk/darknet_lsk.py), HIP backed."""
from glsdet_amd.drone.body import LskCrossYoloBody as YoloBody  # noqa: F401
