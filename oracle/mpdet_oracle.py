"""CPU oracle for the ResNet-50 + FPN + GFLHead / MPHead detectors (SURVEY section 8a rows
A10, A11)  --  TEST INFRASTRUCTURE ONLY (same rules as glsdet_oracle.py: only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import it).

PINNED since round 2, except the NMS.  The reference's implementation of this path lives in ``yolox-ufp/mmdet`` on top of
mmcv-full (pinned ``>=1.3.17,<1.5.0``, yolox-ufp/mmdet/__init__.py:19-27), which is NOT under /root/reference and not
installable here.  tests/golden/make_golden.py --resdet-only therefore loads the reference's OWN files

    ufp/mmdet/models/backbones/resnet.py      (Bottleneck :263-303, ResNet.forward :631-646)
    ufp/mmdet/models/utils/res_layer.py        (downsample :39-61)
    ufp/mmdet/models/necks/fpn.py              (FPN.forward :150-205)
    ufp/mmdet/models/dense_heads/gfl_head.py   (Integral :16-49, forward_single :179-203, _get_bboxes_single :380-471)
    ufp/mmdet/models/dense_heads/mp_head.py    (forward_proxy :105-121, forward_single :123-154)
    ufp/mmdet/models/dense_heads/{anchor_head,base_dense_head,dense_test_mixins}.py (get_bboxes, _bbox_post_process :226-301)
    ufp/mmdet/core/utils/misc.py               (filter_scores_and_topk :119-165, select_single_mlvl)
    ufp/mmdet/core/bbox/transforms.py, coder/distance_point_bbox_coder.py (distance2bbox :153-165)
    ufp/mmdet/core/anchor/anchor_generator.py  (grid_priors)

by file path with stand-ins for the mmcv building blocks they import (build_conv_layer -> nn.Conv2d, build_norm_layer ->
BatchNorm2d, ConvModule = conv (bias only when there is no norm) -> GroupNorm -> ReLU with mmcv's attribute names, Scale =
multiply by a learned scalar, BaseModule / Sequential / registries / init helpers; all listed in make_golden.py) and runs them
on seeded data: tests/golden/resdet_golden.npz (ResNet-50 C2..C5, ResNet-50 + FPN in two configurations, a strided
Bottleneck with downsample) and head_golden.npz (GFLHead and MPHead forward over five levels, Integral, get_bboxes with
with_nms=False at two settings).  This file reproduces all of them (tests/test_resdet_pinned.py: <= 5e-5, candidate lists in
the same order).  What the fixtures cannot cover is ``mmcv.ops.batched_nms`` (a compiled op): the per-class greedy NMS
(IoU > thr suppresses, areas without +1, output in descending score order) stays a restatement -- parity unpinned for that
step, as for torchvision's in the YOLOX path -- and so does the GL-fusion plug-in's WIRING on ResNet (this build's own,
DESIGN.md A12; the plug-in module itself is the pinned Patch_Conv_NonLocal_new).
State-dict key names are mmdet's (torchvision ResNet names for the backbone).

fp16-storage emulation (round 3): inside ``with glsdet_oracle.fp16_storage():`` every tensor is rounded to fp16 where the HIP
path's f16 mode stores fp16 -- the image, every conv weight (the normalised proxies included), the pooled stem output, every
Bottleneck conv after its epilogue (conv3 after the residual add and ReLU), the FPN laterals, each top-down sum, the FPN
outputs, every tower conv before AND after its GroupNorm + ReLU, the MPHead feature -- and stays fp32 where the HIP path
keeps fp32 (BN / GN statistics and affine, accumulators, the regression and proxy logits).  The plug-in's folded
associations reorder its products, so its rounding points are those of glsdet_oracle.patch_conv_nonlocal_new (a model of the
HIP plug-in's storage, not its replica).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from . import glsdet_oracle as _G
from .glsdet_oracle import batched_nms, patch_conv_nonlocal_new

Tensor = torch.Tensor
SD = Dict[str, Tensor]
RESNET_BN_EPS = 1e-5           # nn.BatchNorm2d default (mmcv build_norm_layer(dict(type='BN')))
GN_EPS = 1e-5                  # nn.GroupNorm default
STAGE_BLOCKS = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3)}     # resnet.py arch_settings


def _q(x: Tensor, name: str) -> Tensor:
    """a store point of the HIP path (glsdet_oracle._q: rounds under fp16_storage, feeds TRACE / FORCE)"""
    return _G._q(x, name)


def _w(sd: SD, key: str) -> Tensor:
    """a conv weight as the HIP path holds it (fp16 under fp16_storage)"""
    return _G._qw(sd[key], key[:-len(".weight")])


def _bn(sd: SD, p: str, x: Tensor) -> Tensor:
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"],
                        False, 0.0, RESNET_BN_EPS)


def bottleneck(sd: SD, p: str, x: Tensor, stride: int) -> Tensor:
    """resnet.py:263-303, style='pytorch' (stride on the 3x3): relu(bn3(conv3(relu(bn2(conv2(
    relu(bn1(conv1(x)))))))) + identity) with identity = downsample(x) when present."""
    out = _q(torch.relu(_bn(sd, p + ".bn1", F.conv2d(x, _w(sd, p + ".conv1.weight")))), p + ".conv1")
    out = _q(torch.relu(_bn(sd, p + ".bn2", F.conv2d(out, _w(sd, p + ".conv2.weight"), None, stride, 1))), p + ".conv2")
    out = _bn(sd, p + ".bn3", F.conv2d(out, _w(sd, p + ".conv3.weight")))
    identity = x
    if p + ".downsample.0.weight" in sd:          # res_layer.py:39-61: 1x1 conv (stride s) + BN
        identity = _q(_bn(sd, p + ".downsample.1", F.conv2d(x, _w(sd, p + ".downsample.0.weight"), None, stride)), p + ".downsample")
    return _q(torch.relu(out + identity), p + ".conv3")       # (the HIP epilogue adds the identity before its one store)


def resnet(sd: SD, p: str, x: Tensor, depth: int = 50, out_indices: Sequence[int] = (0, 1, 2, 3)) -> List[Tensor]:
    """resnet.py:631-646: 7x7 s2 conv + BN + ReLU, 3x3 s2 max pool (pad 1), four stages."""
    pre = p + "." if p else ""
    x = torch.relu(_bn(sd, pre + "bn1", F.conv2d(_q(x, "input"), _w(sd, pre + "conv1.weight"), None, 2, 3)))
    x = _q(F.max_pool2d(x, 3, 2, 1), pre + "maxpool")          # (stem + pool are one kernel: one store; max and rounding commute)
    outs = []
    for i, nblocks in enumerate(STAGE_BLOCKS[depth]):
        for j in range(nblocks):
            x = bottleneck(sd, "%slayer%d.%d" % (pre, i + 1, j), x, 2 if (j == 0 and i > 0) else 1)
        if i in out_indices:
            outs.append(x)
    return outs


def _conv_b(sd: SD, p: str, x: Tensor, stride: int = 1, pad: int = 0, store: bool = True, tag: str = "") -> Tensor:
    y = F.conv2d(x, _w(sd, p + ".weight"), sd.get(p + ".bias"), stride, pad)
    return _q(y, p + tag) if store else y


def fpn(sd: SD, p: str, inputs: Sequence[Tensor], start_level: int = 0, num_outs: int = 5,
        add_extra_convs="on_output", relu_before_extra_convs: bool = False) -> List[Tensor]:
    """fpn.py:150-205 (norm_cfg=None, act_cfg=None: every ConvModule is conv + bias only).
    Top-down: laterals[i-1] += interpolate(laterals[i], size=laterals[i-1].shape, 'nearest')."""
    n_lat = len(inputs) - start_level
    lat = [_conv_b(sd, "%s.lateral_convs.%d.conv" % (p, i), inputs[i + start_level]) for i in range(n_lat)]
    for i in range(n_lat - 1, 0, -1):
        lat[i - 1] = _q(lat[i - 1] + F.interpolate(lat[i], size=lat[i - 1].shape[2:], mode="nearest"), "%s.topdown.%d" % (p, i - 1))
    outs = [_conv_b(sd, "%s.fpn_convs.%d.conv" % (p, i), lat[i], 1, 1) for i in range(n_lat)]
    if num_outs > len(outs):
        if not add_extra_convs:
            for _ in range(num_outs - n_lat):
                outs.append(F.max_pool2d(outs[-1], 1, stride=2))
        else:
            src = {"on_input": inputs[-1], "on_lateral": lat[-1], "on_output": outs[-1]}[
                "on_input" if add_extra_convs is True else add_extra_convs]
            outs.append(_conv_b(sd, "%s.fpn_convs.%d.conv" % (p, n_lat), src, 2, 1))
            for i in range(n_lat + 1, num_outs):
                t = torch.relu(outs[-1]) if relu_before_extra_convs else outs[-1]
                outs.append(_conv_b(sd, "%s.fpn_convs.%d.conv" % (p, i), t, 2, 1))
    return outs


def conv_gn_relu(sd: SD, p: str, x: Tensor, groups: int = 32, tag: str = "") -> Tensor:
    """mmcv ConvModule(norm_cfg=GN32): 3x3 conv without bias -> GroupNorm -> ReLU.  tag: suffix of the two store-point
    names (the towers are shared by the levels: '@<level>')."""
    y = _q(F.conv2d(x, _w(sd, p + ".conv.weight"), None, 1, 1), p + ".conv" + tag)
    return _q(torch.relu(F.group_norm(y, groups, sd[p + ".gn.weight"], sd[p + ".gn.bias"], GN_EPS)), p + ".gn" + tag)


def _towers(sd: SD, p: str, x: Tensor, stacked: int, level: int = 0) -> Tuple[Tensor, Tensor]:
    c = r = x
    for i in range(stacked):
        c = conv_gn_relu(sd, "%s.cls_convs.%d" % (p, i), c, tag="@%d" % level)
    for i in range(stacked):
        r = conv_gn_relu(sd, "%s.reg_convs.%d" % (p, i), r, tag="@%d" % level)
    return c, r


def _reg_pred(sd: SD, p: str, r: Tensor, l: int) -> Tensor:
    """gfl_head.py:198-201: Scale_l(gfl_reg(reg_feat)).float().  Under fp16_storage mmcv's Scale is folded into the fp16
    weights, as the HIP path packs them (one rounding of w * s instead of w)."""
    s = sd["%s.scales.%d.scale" % (p, l)]
    if _G._EMU is not None and _G._EMU(p + ".gfl_reg.weight"):
        b = sd.get(p + ".gfl_reg.bias")
        return F.conv2d(r, _G._qw(sd[p + ".gfl_reg.weight"] * s, p + ".gfl_reg"), None if b is None else b * s, 1, 1).float()
    return (_conv_b(sd, p + ".gfl_reg", r, 1, 1, store=False) * s).float()


def gfl_head(sd: SD, p: str, feats: Sequence[Tensor], stacked: int = 4) -> Tuple[List[Tensor], List[Tensor]]:
    """gfl_head.py:179-203; the towers and predictors are SHARED by all levels, Scale is per level."""
    cls, reg = [], []
    for l, x in enumerate(feats):
        c, r = _towers(sd, p, x, stacked, l)
        cls.append(_conv_b(sd, p + ".gfl_cls", c, 1, 1, store=False))
        reg.append(_reg_pred(sd, p, r, l))
    return cls, reg


def forward_proxy(feat: Tensor, proxies: Tensor, proxies_list: Sequence[int], gamma: float) -> Tensor:
    """mp_head.py:105-121: cosine similarity to every proxy; per class a softmax(gamma*sim)
    weighted mean of that class's similarities, times gamma."""
    centers = _G._qw(F.normalize(proxies, p=2, dim=1), "proxies")
    feat = F.normalize(feat, p=2, dim=1)
    sim = feat.matmul(centers.t())
    out, pos = [], 0
    for k in proxies_list:
        sub = sim[:, pos:pos + k]
        out.append(torch.sum(F.softmax(sub * gamma, dim=1) * sub, dim=1)[:, None])
        pos += k
    return torch.cat(out, 1) * gamma


def mp_head(sd: SD, p: str, feats: Sequence[Tensor], proxies_list: Sequence[int], gamma: float = 10.0,
            stacked: int = 4) -> Tuple[List[Tensor], List[Tensor]]:
    """mp_head.py:123-154 (eval branch): cls feature = gfl_cls_conv(cls tower), scored by
    forward_proxy per position."""
    cls, reg = [], []
    for l, x in enumerate(feats):
        c, r = _towers(sd, p, x, stacked, l)
        reg.append(_reg_pred(sd, p, r, l))
        f = _conv_b(sd, p + ".gfl_cls_conv", c, 1, 1, tag="@%d" % l)
        b, ch, h, w = f.shape
        s = forward_proxy(f.permute(0, 2, 3, 1).reshape(-1, ch), sd[p + ".proxies"], proxies_list, gamma)
        cls.append(s.reshape(b, h, w, -1).permute(0, 3, 1, 2).contiguous())
    return cls, reg


def integral(x: Tensor, reg_max: int = 16) -> Tensor:
    """gfl_head.py:16-49: expectation of the softmax over the reg_max+1 bins."""
    x = F.softmax(x.reshape(-1, reg_max + 1), dim=1)
    return F.linear(x, torch.linspace(0, reg_max, reg_max + 1).type_as(x)).reshape(-1, 4)


def gfl_pre_nms(cls_scores: Sequence[Tensor], bbox_preds: Sequence[Tensor], strides: Sequence[int],
                img_shapes: Sequence[Sequence[int]], score_thr: float, nms_pre: int, scale_factors=None,
                reg_max: int = 16) -> List[Tuple[Tensor, Tensor, Tensor]]:
    """base_dense_head.get_bboxes -> gfl_head._get_bboxes_single :380-471 -> _bbox_post_process :226-301 with
    with_nms=False.  Per image and level: sigmoid scores, (position, class) pairs with score > score_thr, the nms_pre
    best of them (filter_scores_and_topk, misc.py:119-165; ties keep the lower flat index first = stable sort),
    Integral * stride, distance2bbox from the anchor centre (x*stride, y*stride) clamped to img_shape; levels
    concatenated, / scale_factor when given.  -> list[img] of (boxes [k,4], scores [k], labels [k]).
    PINNED by tests/golden/head_golden.npz (the reference's own get_bboxes with its AnchorGenerator, bbox coder and
    filter_scores_and_topk)."""
    nc = cls_scores[0].shape[1]
    results = []
    for b in range(cls_scores[0].shape[0]):
        boxes, scores, labels = [], [], []
        for cls, reg, s in zip(cls_scores, bbox_preds, strides):
            h, w = cls.shape[-2:]
            dist = integral(reg[b].permute(1, 2, 0), reg_max) * s
            sc = cls[b].permute(1, 2, 0).reshape(-1, nc).sigmoid()
            valid = sc > score_thr
            vs = sc[valid]
            vidx = torch.nonzero(valid)
            order = torch.sort(vs, descending=True, stable=True)[1][:min(nms_pre, vs.numel())]
            keep, lab = vidx[order].unbind(1)
            gy, gx = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
            cx, cy = (gx.flatten() * s).float()[keep], (gy.flatten() * s).float()[keep]
            d = dist[keep]
            bx = torch.stack((cx - d[:, 0], cy - d[:, 1], cx + d[:, 2], cy + d[:, 3]), -1)
            bx[:, 0::2] = bx[:, 0::2].clamp(min=0, max=img_shapes[b][1])
            bx[:, 1::2] = bx[:, 1::2].clamp(min=0, max=img_shapes[b][0])
            boxes.append(bx), scores.append(vs[order]), labels.append(lab)
        boxes, scores, labels = torch.cat(boxes), torch.cat(scores), torch.cat(labels)
        if scale_factors is not None:
            boxes = boxes / torch.as_tensor(np.asarray(scale_factors[b], np.float32))
        results.append((boxes, scores, labels))
    return results


def gfl_get_bboxes(cls_scores: Sequence[Tensor], bbox_preds: Sequence[Tensor], strides: Sequence[int],
                   img_shapes: Sequence[Sequence[int]], score_thr: float, nms_pre: int, iou_thr: float,
                   max_per_img: int, scale_factors=None, reg_max: int = 16):
    """gfl_pre_nms, then per-class NMS (mmcv.ops.batched_nms -- a compiled op, restated: parity unpinned) and the first
    max_per_img.  -> list[img] of (dets ndarray(n,5) x1,y1,x2,y2,score ; labels ndarray(n) int64)."""
    results = []
    for boxes, scores, labels in gfl_pre_nms(cls_scores, bbox_preds, strides, img_shapes, score_thr, nms_pre, scale_factors, reg_max):
        if boxes.numel() == 0:
            results.append((np.zeros((0, 5), np.float32), np.zeros((0,), np.int64)))
            continue
        bn, sn, ln = boxes.numpy(), scores.numpy(), labels.numpy()
        keep = batched_nms(bn, sn, ln.astype(np.float32), iou_thr)[:max_per_img]
        results.append((np.concatenate([bn[keep], sn[keep][:, None]], 1).astype(np.float32), ln[keep]))
    return results


# ------------------------------------------------------------------------------- GL-fusion plug-in (BASELINE config 3)
def gl_fusion_inputs(sd: SD, p: str, stages: Sequence[Tensor]) -> List[Tensor]:
    """`GLFusionFPN` (authored by this build; the reference never wires GL-fusion onto a ResNet):
    every backbone output i that has a plug-in `<p>.gl_fusion.<i>` becomes  feat + Patch_Conv_NonLocal_new(feat)
    -- the residual of drone/models/new/yolox10.py:262-266 -- before the FPN laterals.  The block itself is the
    pinned restatement of the reference class (glsdet_oracle.patch_conv_nonlocal_new, golden attention_golden.npz)."""
    out = list(stages)
    for i, f in enumerate(stages):
        q = "%s.gl_fusion.%d" % (p, i)
        if q + ".feat_patchconv_lt_nonlocal.theta.weight" in sd:
            out[i] = _q(f + patch_conv_nonlocal_new(sd, q, f), q)
    return out


# ------------------------------------------------------------------------------- detectors
def gfl_forward(sd: SD, x: Tensor, start_level: int = 1, num_outs: int = 5, add_extra_convs="on_output"):
    """SingleStageDetector.extract_feat + bbox_head (single_stage.py:41-60): GFL r50-FPN."""
    feats = fpn(sd, "neck", gl_fusion_inputs(sd, "neck", resnet(sd, "backbone", x)), start_level, num_outs, add_extra_convs)
    return gfl_head(sd, "bbox_head", feats)


def mpdet_forward(sd: SD, x: Tensor, proxies_list: Sequence[int], gamma: float = 10.0, start_level: int = 1,
                  num_outs: int = 5, add_extra_convs="on_output", gl_fusion: Optional[bool] = None):
    """MPDet (mpdet.py:9-18) = SingleStageDetector with MPHead; with a GLFusionFPN neck when the state_dict holds
    neck.gl_fusion.* (gl_fusion=True asserts that it does)."""
    stages = gl_fusion_inputs(sd, "neck", resnet(sd, "backbone", x))
    if gl_fusion:
        assert any(k.startswith("neck.gl_fusion.") for k in sd), "state_dict has no GL-fusion plug-in"
    feats = fpn(sd, "neck", stages, start_level, num_outs, add_extra_convs)
    return mp_head(sd, "bbox_head", feats, proxies_list, gamma)
