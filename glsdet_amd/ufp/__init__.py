"""Second stage of UFPMP-Det (BASELINE config 5): unified foreground
packing of the coarse detections, mosaic compositing on the device, back-mapping and merge NMS of
the fine detections."""
from .packing import unified_foreground_packing  # noqa: F401
from .stage2 import TwoStagePipeline, UfpSecondStage, two_stage_detect  # noqa: F401,E402
