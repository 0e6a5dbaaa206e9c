"""Drop-in root for the reference's `yolox-drone` tree.

Put this directory first on sys.path and the reference's harness (yolo.py: importlib on a
config path string -> `module.YoloBody(num_classes, phi)`; `models.core.utils_bbox`) resolves
to the HIP-backed twins below:

    models/base/yolox.py                              YoloBody            (YOLOX)
    models/block/non_local/yolo_patch_nonlocal_plus.py YoloBody           (YOLOX + GL-fusion neck)
    models/core/utils_bbox.py                         decode_outputs, non_max_suppression,
                                                      yolo_correct_boxes
"""
