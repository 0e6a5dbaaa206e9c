"""mmdet-flavour model surface (ufp/mmdet/models): the classes a config's `type=` names
resolve to -- CSPDarknet, YOLOXPAFPN, YOLOXHead, YOLOX -- plus GLFusionPAFPN, this build's
registry name for the Global-Local fusion neck (the reference wires GL fusion only in its
plain-PyTorch tree).  Same constructor arguments, same state_dict names
(backbone.stage1.0.conv.weight, neck.top_down_blocks.0.main_conv..., bbox_head.multi_level_*),
same call convention and result format; the arithmetic is the libglsdet_hip plan of
glsdet_amd.detector (the mmdet names are mapped onto the drone names).

Reference: ufp/mmdet/models/backbones/csp_darknet.py, necks/yolox_pafpn.py,
dense_heads/yolox_head.py, detectors/{base,single_stage}.py, core/bbox/transforms.py:116-133.
"""
from __future__ import annotations

import re
from collections import OrderedDict
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn as nn

from ..arch import _Table
from ..detector import HipDetector
from ..drone.body import TableModule, _autotune
from .registry import BACKBONES, DETECTORS, HEADS, NECKS, ConfigDict, build_backbone, build_head, build_neck

# ----------------------------------------------------------------------------- name map
_CSP = (("main_conv", "conv1"), ("short_conv", "conv2"), ("final_conv", "conv3"), ("blocks", "m"))
_NECK = (("reduce_layers.0", "lateral_conv0"), ("reduce_layers.1", "reduce_conv1"),
         ("top_down_blocks.0", "C3_p4"), ("top_down_blocks.1", "C3_p3"),
         ("downsamples.0", "bu_conv2"), ("downsamples.1", "bu_conv1"),
         ("bottom_up_blocks.0", "C3_n3"), ("bottom_up_blocks.1", "C3_n4"))
_HEAD = (("multi_level_cls_convs", "cls_convs"), ("multi_level_reg_convs", "reg_convs"),
         ("multi_level_conv_cls", "cls_preds"), ("multi_level_conv_reg", "reg_preds"),
         ("multi_level_conv_obj", "obj_preds"))


_DW = (("depthwise_conv", "dconv"), ("pointwise_conv", "pconv"))     # mmcv DepthwiseSeparableConvModule -> drone DWConv (baseConv.py:22-30)


def _csp_inner(k: str) -> str:
    for a, b in _CSP + _DW:
        k = re.sub(r"(^|\.)%s(\.|$)" % a, r"\g<1>%s\g<2>" % b, k)
    return k


def _dw(k: str) -> str:
    for a, b in _DW:
        k = re.sub(r"(^|\.)%s(\.|$)" % a, r"\g<1>%s\g<2>" % b, k)
    return k


def mmdet_to_drone_key(k: str) -> str:
    """detector-level key (backbone.* / neck.* / bbox_head.*) -> drone YoloBody key."""
    if k.startswith("backbone."):
        r = k[len("backbone."):]
        m = re.match(r"stage(\d)\.(.*)", r)
        if m:
            r = "dark%d.%s" % (int(m.group(1)) + 1, _csp_inner(m.group(2)))
        return "backbone.backbone." + r
    if k.startswith("neck."):
        r = k[len("neck."):]
        m = re.match(r"out_convs\.(\d)\.(.*)", r)
        if m:
            return "head.stems.%s.%s" % (m.group(1), m.group(2))
        for a, b in _NECK:
            if r.startswith(a + "."):
                return "backbone.%s.%s" % (b, _csp_inner(r[len(a) + 1:]))
        return "backbone." + r                       # GL modules keep their names
    if k.startswith("bbox_head."):
        r = k[len("bbox_head."):]
        for a, b in _HEAD:
            if r.startswith(a + "."):
                return "head.%s.%s" % (b, _dw(r[len(a) + 1:]))
    raise KeyError(k)


# ----------------------------------------------------------------------------- modules
_NORM = dict(type="BN", momentum=0.03, eps=0.001)
_ACT = dict(type="Swish")


def _check_cfgs(conv_cfg, norm_cfg, act_cfg, use_depthwise):
    if conv_cfg is not None:
        raise NotImplementedError("conv_cfg other than plain Conv2d is not lowered")
    if dict(norm_cfg).get("type") != "BN" or abs(dict(norm_cfg).get("eps", 1e-5) - 1e-3) > 1e-12:
        raise NotImplementedError("only BN(eps=1e-3) is folded by the HIP path, got %r" % (norm_cfg,))
    if dict(act_cfg).get("type") != "Swish":
        raise NotImplementedError("only Swish/SiLU activations are lowered, got %r" % (act_cfg,))
    # use_depthwise=True (configs/yolox/yolox_nano_8x8_300e_coco.py): mmcv's DepthwiseSeparableConvModule = depthwise kxk ConvModule
    # + pointwise 1x1 ConvModule, each conv + BN + act -- the drone tree's DWConv, lowered by glsdet_dwconv2d + a 1x1 (nets.py)


def _mm_conv(t: _Table, p: str, cin: int, cout: int, k: int, depthwise: bool):
    """ConvModule, or DepthwiseSeparableConvModule with its registration order (depthwise_conv, then pointwise_conv)."""
    if depthwise:
        t.conv_bn(p + ".depthwise_conv", cin, cin, k, groups=cin)
        t.conv_bn(p + ".pointwise_conv", cin, cout, 1)
    else:
        t.conv_bn(p, cin, cout, k)


def _mm_csp(t: _Table, p: str, cin: int, cout: int, n: int, depthwise: bool = False):
    hid = int(cout * 0.5)
    t.conv_bn(p + ".main_conv", cin, hid, 1)
    t.conv_bn(p + ".short_conv", cin, hid, 1)
    t.conv_bn(p + ".final_conv", 2 * hid, cout, 1)
    for i in range(n):
        t.conv_bn("%s.blocks.%d.conv1" % (p, i), hid, hid, 1)
        _mm_conv(t, "%s.blocks.%d.conv2" % (p, i), hid, hid, 3, depthwise)       # utils/csp_layer.py:44-60


@BACKBONES.register_module()
class CSPDarknet(TableModule):
    """ufp/mmdet/models/backbones/csp_darknet.py:123-284 (P5 arch)."""
    arch_settings = {"P5": [[64, 128, 3, True, False], [128, 256, 9, True, False],
                            [256, 512, 9, True, False], [512, 1024, 3, False, True]]}

    def __init__(self, arch="P5", deepen_factor=1.0, widen_factor=1.0, out_indices=(2, 3, 4), frozen_stages=-1,
                 use_depthwise=False, arch_ovewrite=None, spp_kernal_sizes=(5, 9, 13), conv_cfg=None,
                 norm_cfg=_NORM, act_cfg=_ACT, norm_eval=False, init_cfg=None):
        super().__init__()
        if arch not in self.arch_settings or arch_ovewrite:
            raise NotImplementedError("only the P5 CSPDarknet is lowered")
        setting = self.arch_settings[arch]
        assert set(out_indices).issubset(i for i in range(len(setting) + 1))
        if frozen_stages not in range(-1, len(setting) + 1):
            raise ValueError("frozen_stages must be in range(-1, len(arch_setting) + 1). But received %s" % frozen_stages)
        _check_cfgs(conv_cfg, norm_cfg, act_cfg, use_depthwise)
        if tuple(spp_kernal_sizes) != (5, 9, 13) or tuple(out_indices) != (2, 3, 4):
            raise NotImplementedError("lowered for spp_kernal_sizes=(5,9,13), out_indices=(2,3,4)")
        self.out_indices, self.deepen_factor, self.widen_factor = tuple(out_indices), deepen_factor, widen_factor
        t = _Table()
        t.conv_bn("stem.conv", 12, int(setting[0][0] * widen_factor), 3)
        for i, (cin, cout, nb, _add, spp) in enumerate(setting):
            cin, cout = int(cin * widen_factor), int(cout * widen_factor)
            nb = max(round(nb * deepen_factor), 1)
            _mm_conv(t, "stage%d.0" % (i + 1), cin, cout, 3, use_depthwise)      # backbones/csp_darknet.py:212,232-240
            j = 1
            if spp:
                t.conv_bn("stage%d.1.conv1" % (i + 1), cout, cout // 2, 1)
                t.conv_bn("stage%d.1.conv2" % (i + 1), cout // 2 * 4, cout, 1)
                j = 2
            _mm_csp(t, "stage%d.%d" % (i + 1, j), cout, cout, nb, use_depthwise)
        self.out_channels = [int(s[1] * widen_factor) for s in setting][1:]
        self._init_table(t)


def _pafpn_table(in_channels: Sequence[int], out_channels: int, n: int, gl: bool, dw: bool = False) -> _Table:
    if len(in_channels) != 3:
        raise NotImplementedError("lowered for three pyramid levels")
    c = list(in_channels)
    t = _Table()
    extra = 1 if gl else 0
    # registration order of the reference's YOLOXPAFPN.__init__ (necks/yolox_pafpn.py:60-118: the two ModuleLists of the
    # top-down path, then the two of the bottom-up path) -- pinned by tests/test_mmdet_pinned.py against the state_dict of
    # the reference's own module (round 1 interleaved them, which only the ORDER of state_dict() shows)
    t.conv_bn("reduce_layers.0", c[2], c[1], 1)
    t.conv_bn("reduce_layers.1", c[1], c[0], 1)
    _mm_csp(t, "top_down_blocks.0", (2 + extra) * c[1], c[1], n, dw)
    _mm_csp(t, "top_down_blocks.1", 2 * c[0], c[0], n, dw)
    _mm_conv(t, "downsamples.0", c[0], c[0], 3, dw)                          # necks/yolox_pafpn.py:55,84-93
    _mm_conv(t, "downsamples.1", c[1], c[1], 3, dw)
    _mm_csp(t, "bottom_up_blocks.0", (2 + extra) * c[0], c[1], n, dw)
    _mm_csp(t, "bottom_up_blocks.1", 2 * c[1], c[2], n, dw)
    for i in range(3):
        t.conv_bn("out_convs.%d" % i, c[i], out_channels, 1)
    if gl:
        t.plain("P3_Identity.conv", c[0], c[0], 7)
        t.plain("P4_Identity.conv", c[1], c[1], 5)
        t.plain("P5_Identity.conv", c[2], c[2], 3)
        t.patch_conv("Patch_conv_feat1", c[0], c[1], True)
        t.patch_conv("Patch_conv_feat2", c[1], c[0], False)
    return t


@NECKS.register_module()
class YOLOXPAFPN(TableModule):
    """ufp/mmdet/models/necks/yolox_pafpn.py:13-156."""
    gl = False

    def __init__(self, in_channels, out_channels, num_csp_blocks=3, use_depthwise=False,
                 upsample_cfg=dict(scale_factor=2, mode="nearest"), conv_cfg=None, norm_cfg=_NORM, act_cfg=_ACT,
                 init_cfg=None):
        super().__init__()
        _check_cfgs(conv_cfg, norm_cfg, act_cfg, use_depthwise)
        if dict(upsample_cfg) != dict(scale_factor=2, mode="nearest"):
            raise NotImplementedError("only nearest x2 upsampling is lowered")
        self.in_channels, self.out_channels = list(in_channels), out_channels
        self._init_table(_pafpn_table(in_channels, out_channels, num_csp_blocks, self.gl, bool(use_depthwise)))


@NECKS.register_module()
class GLFusionPAFPN(YOLOXPAFPN):
    """YOLOXPAFPN + the Global-Local fusion modules of
    drone/models/block/non_local/yolo_patch_nonlocal_plus.py:94-247 (Patch_Conv_NonLocal on
    the stride-8 map, Patch_Conv on the stride-16 map, Identity convs on P3/P4/P5)."""
    gl = True


@HEADS.register_module()
class YOLOXHead(TableModule):
    """ufp/mmdet/models/dense_heads/yolox_head.py:24-322 (inference part)."""

    def __init__(self, num_classes, in_channels, feat_channels=256, stacked_convs=2, strides=[8, 16, 32],
                 use_depthwise=False, dcn_on_last_conv=False, conv_bias="auto", conv_cfg=None, norm_cfg=_NORM,
                 act_cfg=_ACT, loss_cls=None, loss_bbox=None, loss_obj=None, loss_l1=None, train_cfg=None,
                 test_cfg=None, init_cfg=None):
        super().__init__()
        assert conv_bias == "auto" or isinstance(conv_bias, bool)
        _check_cfgs(conv_cfg, norm_cfg, act_cfg, use_depthwise)
        if stacked_convs != 2 or list(strides) != [8, 16, 32] or dcn_on_last_conv or in_channels != feat_channels:
            raise NotImplementedError("lowered for stacked_convs=2, strides=[8,16,32], in_channels == feat_channels")
        self.num_classes = self.cls_out_channels = num_classes
        self.in_channels, self.feat_channels, self.strides = in_channels, feat_channels, list(strides)
        self.test_cfg = ConfigDict(test_cfg) if test_cfg is not None else None
        self.train_cfg = train_cfg
        t = _Table()
        f = feat_channels
        for name in ("multi_level_cls_convs", "multi_level_reg_convs"):
            for i in range(3):
                for j in range(2):
                    _mm_conv(t, "%s.%d.%d" % (name, i, j), f, f, 3, bool(use_depthwise))    # dense_heads/yolox_head.py:146-165
        for name, co in (("multi_level_conv_cls", num_classes), ("multi_level_conv_reg", 4), ("multi_level_conv_obj", 1)):
            for i in range(3):
                t.plain("%s.%d" % (name, i), f, co, 1)
        self._init_table(t)


def bbox2result(dets: np.ndarray, num_classes: int) -> List[np.ndarray]:
    """ufp/mmdet/core/bbox/transforms.py:116-133 on our (k,7) rows [x1,y1,x2,y2,obj,cls_conf,cls]."""
    if dets.shape[0] == 0:
        return [np.zeros((0, 5), dtype=np.float32) for _ in range(num_classes)]
    b5 = np.concatenate([dets[:, :4], (dets[:, 4] * dets[:, 5])[:, None]], 1).astype(np.float32)
    labels = dets[:, 6].astype(np.int64)
    return [b5[labels == i, :] for i in range(num_classes)]


@DETECTORS.register_module()
class YOLOX(nn.Module):
    """Single-stage detector (ufp/mmdet/models/detectors/single_stage.py:19-108 with the
    YOLOX config keys of configs/yolox/yolox_s_8x8_300e_coco.py:7-10; the reference checkout
    lacks detectors/yolox.py, F3).  `model(return_loss=False, rescale=True, img=[Tensor],
    img_metas=[[dict]])` -> list[img] of list[class] of ndarray(n,5)."""

    def __init__(self, backbone, neck=None, bbox_head=None, train_cfg=None, test_cfg=None, pretrained=None,
                 input_size=(640, 640), size_multiplier=32, random_size_range=(15, 25), random_size_interval=10,
                 init_cfg=None, hip_dtype="f16"):
        super().__init__()
        self.backbone = build_backbone(backbone)
        self.neck = build_neck(neck) if neck is not None else None
        bbox_head = dict(bbox_head)
        bbox_head.update(train_cfg=train_cfg)
        bbox_head.update(test_cfg=test_cfg)
        self.bbox_head = build_head(bbox_head)
        self.train_cfg, self.test_cfg = train_cfg, ConfigDict(test_cfg) if test_cfg is not None else None
        self.hip_dtype = hip_dtype
        self._det: Optional[HipDetector] = None
        self.CLASSES = None

    @property
    def with_neck(self):
        return self.neck is not None

    def load_state_dict(self, state_dict, strict: bool = True):
        sd = state_dict.get("state_dict", state_dict)          # mmcv checkpoints: {'meta':..., 'state_dict':...}
        parts: Dict[str, dict] = {"backbone": {}, "neck": {}, "bbox_head": {}}
        other = []
        for k, v in sd.items():
            k = k[7:] if k.startswith("module.") else k
            head, _, rest = k.partition(".")
            if head in parts:
                parts[head][rest] = v
            else:
                other.append(k)
        if strict and other:
            raise RuntimeError("unexpected keys: %s" % other[:5])
        for name, sub in parts.items():
            getattr(self, name).load_state_dict(sub, strict=strict)
        self._det = None

    def _detector(self) -> HipDetector:
        if self._det is None:
            sd = OrderedDict((mmdet_to_drone_key(k), v) for k, v in self.state_dict().items())
            self._det = HipDetector("gl" if getattr(self.neck, "gl", False) else "base", sd, dtype=self.hip_dtype, autotune=_autotune())
        return self._det

    # ---- call convention of BaseDetector.forward (ufp/mmdet/models/detectors/base.py:157-175)
    def forward(self, img, img_metas, return_loss=True, **kwargs):
        if return_loss:
            raise NotImplementedError("glsdet_amd implements the inference forward only (return_loss=False)")
        return self.forward_test(img, img_metas, **kwargs)

    def forward_test(self, imgs, img_metas, **kwargs):
        for var, name in [(imgs, "imgs"), (img_metas, "img_metas")]:
            if not isinstance(var, list):
                raise TypeError("{} must be a list, but got {}".format(name, type(var)))
        if len(imgs) != len(img_metas):
            raise ValueError("num of augmentations ({}) != num of image meta ({})".format(len(imgs), len(img_metas)))
        if len(imgs) != 1:
            raise NotImplementedError("test-time augmentation is outside the hot path")
        return self.simple_test(imgs[0], img_metas[0], **kwargs)

    def extract_feat(self, img):
        """backbone + neck -> tuple of [B, out_channels, H_l, W_l] (the neck's out_convs outputs)."""
        det = self._detector()
        img = img.to("cuda", torch.float32)
        c = det.compile(img.shape[0], img.shape[2], img.shape[3])
        det.run(c, img)
        return tuple(v.to_nchw() for v in c.stems)

    def simple_test(self, img, img_metas, rescale=False):
        if self.training:
            raise NotImplementedError("call .eval(): inference only")
        cfg = self.bbox_head.test_cfg or self.test_cfg
        if cfg is None:
            raise ValueError("test_cfg (score_thr, nms.iou_threshold) is required")
        det = self._detector()
        img = img.to("cuda", torch.float32)
        n, _, H, W = img.shape
        post = dict(conf_thres=float(cfg["score_thr"]), nms_thres=float(cfg["nms"]["iou_threshold"]),
                    max_det=H // 8 * (W // 8) + H // 16 * (W // 16) + H // 32 * (W // 32), mode=1, rescale=bool(rescale))
        c = det.compile(n, H, W, post)
        scale = None
        if rescale:
            scale = torch.tensor(np.stack([np.asarray(m["scale_factor"], np.float32).reshape(-1)[:4] for m in img_metas]),
                                 dtype=torch.float32)
        det.run(c, img, scale=scale)
        dets = HipDetector.collect(c)
        return [bbox2result(d, self.bbox_head.num_classes) for d in dets]
