"""mmdet-style surface: `Config.fromfile` -> `build_detector(cfg.model)` / `init_detector`
with the reference's `type=` names, argument names, state_dict names and result format."""
from .registry import (BACKBONES, DETECTORS, HEADS, MODELS, NECKS, Config, ConfigDict, Registry,  # noqa: F401
                       build_backbone, build_detector, build_head, build_neck)
from .models import (CSPDarknet, GLFusionPAFPN, YOLOX, YOLOXHead, YOLOXPAFPN, bbox2result,  # noqa: F401
                     mmdet_to_drone_key)
from .resdet_models import FPN, GFL, GFLHead, MPDet, MPHead, ResNet, SingleStageDetector  # noqa: F401


def init_detector(config, checkpoint=None, device="cuda:0", cfg_options=None):
    """ufp/mmdet/apis/inference.py:17-53: config file (or Config) -> detector in eval mode with
    `.cfg` attached; `checkpoint` = mmcv-style {'state_dict', 'meta'} or bare state_dict file."""
    import torch
    if isinstance(config, str):
        config = Config.fromfile(config)
    elif not isinstance(config, Config):
        raise TypeError("config must be a filename or Config object, but got %s" % type(config))
    if cfg_options is not None:
        config.merge_from_dict(cfg_options)
    model_cfg = dict(config.model)
    model_cfg.pop("pretrained", None)
    model_cfg.pop("train_cfg", None)
    model = build_detector(model_cfg, test_cfg=config.get("test_cfg"))
    if checkpoint is not None:
        ck = torch.load(checkpoint, map_location="cpu", weights_only=True)
        model.load_state_dict(ck)
        meta = ck.get("meta", {}) if isinstance(ck, dict) else {}
        if "CLASSES" in meta:
            model.CLASSES = meta["CLASSES"]
    model.cfg = config
    model.eval()
    return model
