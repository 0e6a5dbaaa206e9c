"""Twin of drone/models/new/yolox6.py (the reference's default `--model-config`,
get_map.py:30): `YoloBody(num_classes, phi)` with the cross-scale decoupled head, HIP backed."""
from glsdet_amd.drone.body import CrossYoloBody as YoloBody  # noqa: F401
