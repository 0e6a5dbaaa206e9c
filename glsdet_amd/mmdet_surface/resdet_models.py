"""mmdet-flavour surface of the ResNet-50 + FPN + GFLHead / MPHead detectors: the classes the
UFPMP-Det configs name (`type='GFL'` coarse detector, `type='MPDet'` fine detector,
ufp/ufpmp_det_eval.py:218-227) with the reference's constructor arguments, state_dict names,
call convention and result format.  Arithmetic = the libglsdet_hip plan of
glsdet_amd.resdet.HipGflDetector.

Reference: ufp/mmdet/models/backbones/resnet.py:306-470, necks/fpn.py:62-148,
dense_heads/gfl_head.py:87-152, dense_heads/mp_head.py:23-91, dense_heads/anchor_head.py,
detectors/{single_stage.py:19-108, mpdet.py, base.py:157-175}, core/bbox/transforms.py:116-133.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn as nn

from ..arch import _Table, fpn_table, gfl_head_table, gl_fusion_table, mp_head_table, resnet_table, RESNET_STAGE_BLOCKS
from ..drone.body import TableModule, _autotune
from ..resdet import HipGflDetector
from .registry import BACKBONES, DETECTORS, HEADS, NECKS, ConfigDict, build_backbone, build_head, build_neck


@BACKBONES.register_module()
class ResNet(TableModule):
    """ufp/mmdet/models/backbones/resnet.py:306-470.  Lowered: depth 50/101, 4 stages, style
    'pytorch', BN, no deep stem / avg_down / DCN / plugins (each rejected with the argument named)."""
    arch_settings = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3)}

    def __init__(self, depth, in_channels=3, stem_channels=None, base_channels=64, num_stages=4, strides=(1, 2, 2, 2),
                 dilations=(1, 1, 1, 1), out_indices=(0, 1, 2, 3), style="pytorch", deep_stem=False, avg_down=False,
                 frozen_stages=-1, conv_cfg=None, norm_cfg=dict(type="BN", requires_grad=True), norm_eval=True, dcn=None,
                 stage_with_dcn=(False, False, False, False), plugins=None, with_cp=False, zero_init_residual=True,
                 pretrained=None, init_cfg=None):
        super().__init__()
        if depth not in (18, 34, 50, 101, 152):
            raise KeyError("invalid depth %s for resnet" % depth)
        if depth not in self.arch_settings:
            raise NotImplementedError("only the Bottleneck ResNets of depth 50 and 101 are lowered")
        assert 1 <= num_stages <= 4 and len(strides) == len(dilations) == num_stages
        assert max(out_indices) < num_stages
        for name, val, ok in (("in_channels", in_channels, 3), ("stem_channels", stem_channels or base_channels, 64),
                              ("base_channels", base_channels, 64), ("num_stages", num_stages, 4),
                              ("strides", tuple(strides), (1, 2, 2, 2)), ("dilations", tuple(dilations), (1, 1, 1, 1)),
                              ("style", style, "pytorch"), ("deep_stem", deep_stem, False), ("avg_down", avg_down, False),
                              ("conv_cfg", conv_cfg, None), ("dcn", dcn, None), ("plugins", plugins, None)):
            if val != ok:
                raise NotImplementedError("ResNet(%s=%r) is not lowered (supported: %r)" % (name, val, ok))
        if (norm_cfg or {}).get("type", "BN") != "BN":
            raise NotImplementedError("ResNet norm_cfg type must be BN")
        self.depth, self.out_indices = depth, tuple(out_indices)
        self.out_channels = [256, 512, 1024, 2048]
        t = _Table()
        resnet_table(t, "x", depth)
        self._init_table(_Table((k[2:], v) for k, v in t.items()))


@NECKS.register_module()
class FPN(TableModule):
    """ufp/mmdet/models/necks/fpn.py:62-148 (norm_cfg=None, act_cfg=None, nearest upsampling)."""

    def __init__(self, in_channels, out_channels, num_outs, start_level=0, end_level=-1, add_extra_convs=False,
                 relu_before_extra_convs=False, no_norm_on_lateral=False, conv_cfg=None, norm_cfg=None, act_cfg=None,
                 upsample_cfg=dict(mode="nearest"), init_cfg=None):
        super().__init__()
        assert isinstance(in_channels, (list, tuple))
        assert isinstance(add_extra_convs, (str, bool))
        if isinstance(add_extra_convs, str):
            assert add_extra_convs in ("on_input", "on_lateral", "on_output")
        if end_level != -1:
            raise NotImplementedError("FPN end_level != -1 is not lowered")
        assert num_outs >= len(in_channels) - start_level
        if conv_cfg or norm_cfg or act_cfg or dict(upsample_cfg) != dict(mode="nearest") or relu_before_extra_convs:
            raise NotImplementedError("FPN is lowered for plain convs, nearest upsampling by size, no relu before extras")
        self.in_channels, self.out_channels, self.num_outs = list(in_channels), out_channels, num_outs
        self.start_level, self.add_extra_convs = start_level, add_extra_convs
        t = _Table()
        fpn_table(t, "x", self.in_channels, out_channels, start_level, num_outs, add_extra_convs)
        self._init_table(_Table((k[2:], v) for k, v in t.items()))


@NECKS.register_module()
class GLFusionFPN(FPN):
    """FPN whose inputs first pass the GL-fusion plug-in: in_i <- in_i + Patch_Conv_NonLocal_new(in_i) on the backbone
    outputs `gl_levels` (default C3, C4, C5), the residual of drone/models/new/yolox10.py:262-266 with the block of
    drone/models/new/Non_local_family.py:208-252 (in = out = C_i, channel_scale = 1, patch_scale = 2).  Authored by this
    build: BASELINE config 3 names "ResNet-50 + GL-fusion + decoupled head", a model the reference never wires up
    (SURVEY F4, App. B).  gl_channel_cat: 'linear' (1x1 conv + bias) or 'non_linear' (BaseConv 3x3 + BN + SiLU), the two
    options of the reference class.  State-dict names: gl_fusion.<i>.feat_patchconv_{lt,lb,rt,rb}_nonlocal.{g,theta,phi,
    conv_out}.{weight,bias}, gl_fusion.<i>.channel_conv.*."""

    def __init__(self, in_channels, out_channels, num_outs, gl_levels=(1, 2, 3), gl_channel_cat="linear", **kwargs):
        super().__init__(in_channels, out_channels, num_outs, **kwargs)
        if gl_channel_cat not in ("linear", "non_linear"):
            raise ValueError("gl_channel_cat must be 'linear' or 'non_linear'")
        self.gl_levels, self.gl_channel_cat = tuple(gl_levels), gl_channel_cat
        assert all(0 <= i < len(self.in_channels) for i in self.gl_levels)
        t = _Table()
        fpn_table(t, "x", self.in_channels, out_channels, self.start_level, num_outs, self.add_extra_convs)
        for i in self.gl_levels:
            gl_fusion_table(t, "x.gl_fusion.%d" % i, self.in_channels[i], gl_channel_cat)
        self._init_table(_Table((k[2:], v) for k, v in t.items()))


class _GflLikeHead(TableModule):
    def _common(self, num_classes, in_channels, feat_channels, stacked_convs, anchor_generator, reg_max, norm_cfg,
                conv_cfg, train_cfg, test_cfg):
        ag = dict(anchor_generator or {})
        self.strides = tuple(int(s if not isinstance(s, (list, tuple)) else s[0]) for s in ag.get("strides", (8, 16, 32, 64, 128)))
        if ag and (list(ag.get("ratios", [1.0])) != [1.0] or ag.get("scales_per_octave", 1) != 1):
            raise AssertionError("anchor free version")            # gfl_head.py:147 (num_anchors == 1)
        if conv_cfg or dict(norm_cfg or {}).get("type") != "GN" or dict(norm_cfg).get("num_groups") != 32:
            raise NotImplementedError("head towers are lowered for conv + GN(32) + ReLU")
        self.num_classes, self.in_channels, self.feat_channels = num_classes, in_channels, feat_channels
        self.stacked_convs, self.reg_max = stacked_convs, reg_max
        self.train_cfg, self.test_cfg = train_cfg, ConfigDict(test_cfg) if test_cfg is not None else None


@HEADS.register_module()
class GFLHead(_GflLikeHead):
    """ufp/mmdet/models/dense_heads/gfl_head.py:87-152."""

    def __init__(self, num_classes, in_channels, stacked_convs=4, feat_channels=256, conv_cfg=None,
                 norm_cfg=dict(type="GN", num_groups=32, requires_grad=True), anchor_generator=None, loss_cls=None,
                 loss_bbox=None, loss_dfl=None, bbox_coder=None, reg_max=16, train_cfg=None, test_cfg=None, init_cfg=None):
        super().__init__()
        self._common(num_classes, in_channels, feat_channels, stacked_convs, anchor_generator, reg_max, norm_cfg, conv_cfg,
                     train_cfg, test_cfg)
        t = _Table()
        gfl_head_table(t, "x", num_classes, in_channels, feat_channels, stacked_convs, reg_max, len(self.strides))
        self._init_table(_Table((k[2:], v) for k, v in t.items()))


@HEADS.register_module()
class MPHead(_GflLikeHead):
    """ufp/mmdet/models/dense_heads/mp_head.py:23-91."""

    def __init__(self, num_words=200, beta=0, gamma=10, proxies_list=[2, 3, 2, 5, 4, 8, 8, 4, 3, 3], **kwargs):
        super().__init__()
        self.num_words, self.beta, self.gamma, self.proxies_list = num_words, beta, gamma, list(proxies_list)
        kw = dict(stacked_convs=4, feat_channels=256, conv_cfg=None, norm_cfg=dict(type="GN", num_groups=32, requires_grad=True),
                  anchor_generator=None, reg_max=16, train_cfg=None, test_cfg=None)
        for k in ("loss_cls", "loss_bbox", "loss_dfl", "bbox_coder", "init_cfg", "loss_op", "loss_emd"):
            kwargs.pop(k, None)
        num_classes, in_channels = kwargs.pop("num_classes"), kwargs.pop("in_channels")
        unknown = set(kwargs) - set(kw)
        if unknown:
            raise TypeError("MPHead got unexpected arguments %s" % sorted(unknown))
        kw.update(kwargs)
        self._common(num_classes, in_channels, kw["feat_channels"], kw["stacked_convs"], kw["anchor_generator"],
                     kw["reg_max"], kw["norm_cfg"], kw["conv_cfg"], kw["train_cfg"], kw["test_cfg"])
        assert self.num_classes == len(self.proxies_list)          # mp_head.py:39
        t = _Table()
        mp_head_table(t, "x", self.proxies_list, in_channels, self.feat_channels, self.stacked_convs, self.reg_max,
                      len(self.strides), num_words)
        self._init_table(_Table((k[2:], v) for k, v in t.items()))


def bbox2result(bboxes: np.ndarray, labels: np.ndarray, num_classes: int) -> List[np.ndarray]:
    """ufp/mmdet/core/bbox/transforms.py:116-133."""
    if bboxes.shape[0] == 0:
        return [np.zeros((0, 5), dtype=np.float32) for _ in range(num_classes)]
    return [bboxes[labels == i, :] for i in range(num_classes)]


@DETECTORS.register_module()
class SingleStageDetector(nn.Module):
    """ufp/mmdet/models/detectors/single_stage.py:19-108 for ResNet + FPN + GFLHead / MPHead.
    `model(return_loss=False, rescale=True, img=[Tensor], img_metas=[[dict]])` ->
    list[img] of list[class] of float32 ndarray (n,5) = x1,y1,x2,y2,score."""
    head_kind = None

    def __init__(self, backbone, neck=None, bbox_head=None, train_cfg=None, test_cfg=None, pretrained=None,
                 init_cfg=None, hip_dtype="f16"):
        super().__init__()
        self.backbone = build_backbone(backbone)
        self.neck = build_neck(neck) if neck is not None else None
        bbox_head = dict(bbox_head)
        bbox_head.update(train_cfg=train_cfg)
        bbox_head.update(test_cfg=test_cfg)
        self.bbox_head = build_head(bbox_head)
        if not isinstance(self.backbone, ResNet) or not isinstance(self.neck, FPN) or \
                not isinstance(self.bbox_head, (GFLHead, MPHead)):
            raise NotImplementedError("%s is lowered for ResNet + FPN + GFLHead/MPHead" % type(self).__name__)
        self.train_cfg, self.test_cfg = train_cfg, ConfigDict(test_cfg) if test_cfg is not None else None
        self.hip_dtype = hip_dtype
        self._det: Optional[HipGflDetector] = None
        self.CLASSES = None

    @property
    def with_neck(self):
        return self.neck is not None

    def load_state_dict(self, state_dict, strict: bool = True):
        sd = state_dict.get("state_dict", state_dict)
        parts: Dict[str, dict] = {"backbone": {}, "neck": {}, "bbox_head": {}}
        other = []
        for k, v in sd.items():
            k = k[7:] if k.startswith("module.") else k
            head, _, rest = k.partition(".")
            (parts[head].__setitem__(rest, v) if head in parts else other.append(k))
        if strict and other:
            raise RuntimeError("unexpected keys: %s" % other[:5])
        for name, sub in parts.items():
            getattr(self, name).load_state_dict(sub, strict=strict)
        self._det = None

    def state_dict(self, *args, destination=None, prefix="", keep_vars=False):
        out = destination if destination is not None else {}
        for name in ("backbone", "neck", "bbox_head"):
            getattr(self, name).state_dict(destination=out, prefix=prefix + name + ".", keep_vars=keep_vars)
        return out

    def _detector(self) -> HipGflDetector:
        if self._det is None:
            h, n = self.bbox_head, self.neck
            kind = "mpdet" if isinstance(h, MPHead) else "gfl"
            cfg = dict(start_level=n.start_level, num_outs=n.num_outs, add_extra_convs=n.add_extra_convs,
                       stacked_convs=h.stacked_convs, strides=h.strides, reg_max=h.reg_max, depth=self.backbone.depth,
                       out_indices=self.backbone.out_indices)
            if kind == "mpdet":
                cfg.update(proxies_list=tuple(h.proxies_list), gamma=float(h.gamma))
            cfg.update(gl_levels=list(n.gl_levels) if isinstance(n, GLFusionFPN) else [])
            self._det = HipGflDetector(kind, self.state_dict(), dtype=self.hip_dtype, autotune=_autotune(), **cfg)
        return self._det

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

    def simple_test(self, img, img_metas, rescale=False):
        if self.training:
            raise NotImplementedError("call .eval(): inference only")
        cfg = self.bbox_head.test_cfg or self.test_cfg
        if cfg is None:
            raise ValueError("test_cfg (score_thr, nms, nms_pre, max_per_img) is required")
        det = self._detector()
        res = det.detect(img.to("cuda", torch.float32), score_thr=float(cfg["score_thr"]),
                         iou_thr=float(cfg["nms"]["iou_threshold"]), nms_pre=int(cfg.get("nms_pre", -1)) if
                         int(cfg.get("nms_pre", -1)) > 0 else 1000, max_per_img=int(cfg.get("max_per_img", 100)),
                         img_shapes=[m["img_shape"] for m in img_metas],
                         scale_factors=[np.asarray(m["scale_factor"], np.float32).reshape(-1)[:4] for m in img_metas]
                         if rescale else None)
        return [bbox2result(d, l, self.bbox_head.num_classes) for d, l in res]


@DETECTORS.register_module()
class GFL(SingleStageDetector):
    """ufp/mmdet/models/detectors/gfl.py (SingleStageDetector with a GFLHead)."""


@DETECTORS.register_module()
class MPDet(SingleStageDetector):
    """ufp/mmdet/models/detectors/mpdet.py:9-18."""
