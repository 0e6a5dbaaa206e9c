"""CPU oracle for the GLSDet detection forward pass  --  TEST INFRASTRUCTURE ONLY.

This file is the checker, never the product: only ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py`` may import it.  The product path
(``glsdet_amd``) never touches it and fails loudly when its HIP library is missing.

It is a plain fp32 restatement (torch CPU, functional, NCHW) of the reference's
algorithm, driven directly by a state_dict that uses the *reference's* key names, so a
reference checkpoint and a reference-generated golden vector can both be fed to it.
Every function cites the reference file:line it follows (paths relative to
``/root/reference``; ``drone/`` = ``yolox-drone/``).

Pinning: ``tests/test_oracle_golden.py`` checks this file against golden vectors that
``tests/golden/make_golden.py`` produced by importing the reference itself
(``yolox-drone`` tree, CPU) in the build container.  NMS is the one exception: the
reference calls ``torchvision.ops.boxes.batched_nms`` which is not importable here, so
``batched_nms`` below is restated from torchvision's documented semantics and is
"parity unpinned" (see DESIGN.md).

fp16-storage emulation (``with fp16_storage(): ...``): the same restatement with every
tensor rounded to fp16 exactly where the HIP path's benchmarked mode stores fp16 -- the
input image, every conv weight, every conv / non-local output AFTER its fused epilogue
(BN scale+bias, activation, residual add) -- and fp32 everywhere the HIP path keeps fp32
(accumulators, BN scale/bias, the non-local Gram / fold matrices and its conv_out weights,
the head logits, decode, NMS).  It separates kernel error (HIP f16 vs this) from the
conditioning of the net (this vs the fp32 reference); ``fp16_storage(select=...)`` rounds
only the named tensors and gives the per-layer attribution table of DESIGN.md.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]

# drone/models/base/yolox.py:240-241  (depth / width tables of YoloBody)
DEPTH = {"nano": 0.33, "tiny": 0.33, "s": 0.33, "m": 0.67, "l": 1.00, "x": 1.33}
WIDTH = {"nano": 0.25, "tiny": 0.375, "s": 0.50, "m": 0.75, "l": 1.00, "x": 1.25}
BN_EPS = 1e-3  # drone/models/base/baseConv.py:12

# --------------------------------------------------------------------------- fp16-storage emulation
_EMU = None          # None: plain fp32.  Else callable(name) -> bool: round the tensor called `name` to fp16
TRACE = None         # optional dict: name -> the (possibly rounded) tensor every conv-like op produced
CALIBRATE = None     # optional numpy Generator: every BaseConv overwrites ITS running_mean / running_var in the state_dict
#                      by perturbed batch statistics of what it is fed in this very forward (calibrate_bn below)
FORCE = None         # optional dict: name -> tensor that REPLACES the op's output after it was traced
#                      ("teacher forcing": with the HIP path's own stored tensors here, every op of the
#                      oracle consumes exactly what the corresponding kernel consumed, so a difference at one
#                      store point is that kernel's own error and nothing propagates to the next)


class fp16_storage:
    """Context manager: emulate the HIP path's fp16 storage (see the module docstring).
    select: None (round everything the HIP path rounds) or callable(name) -> bool, where name is
    'input', '<prefix>.weight' for a conv weight or '<prefix>' for the output of the op at <prefix>."""

    def __init__(self, select=None):
        self.select = select if select is not None else (lambda name: True)

    def __enter__(self):
        global _EMU
        self._old, _EMU = _EMU, self.select
        return self

    def __exit__(self, *exc):
        global _EMU
        _EMU = self._old
        return False


def _q(x: Tensor, name: str) -> Tensor:
    """Store point of the HIP path: round to fp16 (round-to-nearest-even, as v_cvt_f16_f32) when emulating."""
    if _EMU is not None and x.dtype == torch.float32 and _EMU(name):
        x = x.half().float()
    if TRACE is not None:
        TRACE[name] = x
    if FORCE is not None and name in FORCE:
        f = FORCE[name]
        assert f.shape == x.shape, (name, tuple(f.shape), tuple(x.shape))
        x = f.to(x.dtype)
    return x


def _qw(w: Tensor, name: str) -> Tensor:
    if _EMU is not None and w.dtype == torch.float32 and _EMU(name + ".weight"):
        return w.half().float()
    return w


# --------------------------------------------------------------------------- primitives
def _act(x: Tensor, kind: str) -> Tensor:
    # drone/models/base/activation.py:4-17
    if kind == "silu":
        return x * torch.sigmoid(x)
    if kind == "relu":
        return torch.relu(x)
    if kind == "lrelu":
        return F.leaky_relu(x, 0.1)
    if kind == "none":
        return x
    raise AttributeError("Unsupported act type: {}".format(kind))


def base_conv(sd: SD, p: str, x: Tensor, stride: int = 1, act: str = "silu", res: Optional[Tensor] = None) -> Tensor:
    """act(bn(conv(x))) [+ res], conv without bias, pad=(k-1)//2, eval-mode BN.
    drone/models/base/baseConv.py:6-16.  groups is inferred from the weight shape.
    `res` is the Bottleneck shortcut (darknet.py:61-62): the HIP path adds it in the conv epilogue
    before the one fp16 store, so the emulation rounds after the add."""
    w = _qw(sd[p + ".conv.weight"], p + ".conv")
    k = w.shape[-1]
    groups = x.shape[1] // w.shape[1]
    y = F.conv2d(x, w, None, stride, (k - 1) // 2, 1, groups)
    if CALIBRATE is not None:
        _calibrate(sd, p, y)
    y = F.batch_norm(y, sd[p + ".bn.running_mean"], sd[p + ".bn.running_var"],
                     sd[p + ".bn.weight"], sd[p + ".bn.bias"], False, 0.0, BN_EPS)
    y = _act(y, act)
    if res is not None:
        y = y + res
    return _q(y, p)


def _calibrate(sd: SD, p: str, y: Tensor) -> None:
    """The recipe of tests/golden/make_golden.calibrate_bn (data preparation, not part of the reference's algorithm):
    running stats := perturbed batch statistics of the BN's input, variance floored so that no channel amplifies by
    more than ~1.4x -- the signal of a random-weight net then neither dies nor explodes over ~80 layers."""
    var = y.var((0, 2, 3), unbiased=False)
    var = var + 0.5 * var.mean() + 1e-4
    mean = y.mean((0, 2, 3))
    c = mean.numel()
    dt = y.dtype
    sd[p + ".bn.running_var"] = var * torch.from_numpy(CALIBRATE.uniform(0.8, 1.25, c).astype(np.float32)).to(dt)
    sd[p + ".bn.running_mean"] = mean + var.sqrt() * torch.from_numpy((0.1 * CALIBRATE.standard_normal(c)).astype(np.float32)).to(dt)


def calibrate_bn(forward, sd: SD, x: Tensor, seed: int = 0) -> SD:
    """-> a copy of sd whose BN running statistics are calibrated on x by ONE forward (each BN on what its already
    calibrated upstream produces).  Test data preparation: the goldens' stored statistics were calibrated at 128 x 160;
    at the benchmark's 800 x 1344 the same weights are badly conditioned (the fp32 oracle sits 1e-2 from its own fp64
    evaluation), so the full-size parity tests calibrate at their own size."""
    global CALIBRATE
    out = dict(sd)
    CALIBRATE = np.random.default_rng([seed, 0xB17])
    try:
        with torch.no_grad():
            forward(out, x)
    finally:
        CALIBRATE = None
    return out


def dw_conv(sd: SD, p: str, x: Tensor, stride: int = 1, act: str = "silu", res: Optional[Tensor] = None) -> Tensor:
    # drone/models/base/baseConv.py:22-30  depthwise kxk then pointwise 1x1
    return base_conv(sd, p + ".pconv", base_conv(sd, p + ".dconv", x, stride, act), 1, act, res)


def any_conv(sd: SD, p: str, x: Tensor, stride: int = 1, act: str = "silu", res: Optional[Tensor] = None) -> Tensor:
    """BaseConv or DWConv depending on what the checkpoint holds at prefix p."""
    if p + ".dconv.conv.weight" in sd:
        return dw_conv(sd, p, x, stride, act, res)
    return base_conv(sd, p, x, stride, act, res)


def plain_conv(sd: SD, p: str, x: Tensor, stride: int = 1, pad: int = 0, store: bool = True,
               post=None, weights_fp32: bool = False) -> Tensor:
    """nn.Conv2d with bias (predictors, non-local projections, Identity_Conv).  `post` = what the HIP
    path fuses into this conv's epilogue before its store (GELU, sigmoid, a residual add);
    store=False: the output stays fp32 in the HIP path (head logits); weights_fp32: so do the weights."""
    w = sd[p + ".weight"] if weights_fp32 else _qw(sd[p + ".weight"], p)
    y = F.conv2d(x, w, sd.get(p + ".bias"), stride, pad)
    if post is not None:
        y = post(y)
    return _q(y, p) if store else y


# --------------------------------------------------------------------------- backbone
def focus(sd: SD, p: str, x: Tensor) -> Tensor:
    # drone/models/base/darknet.py:15-21  order: TL, BL, TR, BR
    x = _q(x, "input")                                  # glsdet_focus_pack stores the packed image in fp16
    tl, bl = x[..., 0::2, 0::2], x[..., 1::2, 0::2]
    tr, br = x[..., 0::2, 1::2], x[..., 1::2, 1::2]
    return base_conv(sd, p + ".conv", torch.cat((tl, bl, tr, br), 1))


def spp_bottleneck(sd: SD, p: str, x: Tensor, ks: Sequence[int] = (5, 9, 13)) -> Tensor:
    # drone/models/base/darknet.py:24-37
    x = base_conv(sd, p + ".conv1", x)
    x = torch.cat([x] + [F.max_pool2d(x, k, 1, k // 2) for k in ks], 1)
    return base_conv(sd, p + ".conv2", x)


def bottleneck(sd: SD, p: str, x: Tensor, shortcut: bool) -> Tensor:
    # drone/models/base/darknet.py:43-63
    add = shortcut and sd[p + ".conv1.conv.weight"].shape[1] == \
        (sd.get(p + ".conv2.conv.weight", sd.get(p + ".conv2.pconv.conv.weight"))).shape[0]
    return any_conv(sd, p + ".conv2", base_conv(sd, p + ".conv1", x), res=x if add else None)


def csp_layer(sd: SD, p: str, x: Tensor, shortcut: bool = True) -> Tensor:
    # drone/models/base/darknet.py:66-112 ; number of bottlenecks read off the checkpoint
    a = base_conv(sd, p + ".conv1", x)
    b = base_conv(sd, p + ".conv2", x)
    i = 0
    while "{}.m.{}.conv1.conv.weight".format(p, i) in sd:
        a = bottleneck(sd, "{}.m.{}".format(p, i), a, shortcut)
        i += 1
    return base_conv(sd, p + ".conv3", torch.cat((a, b), 1))


def csp_darknet(sd: SD, p: str, x: Tensor) -> Dict[str, Tensor]:
    # drone/models/base/darknet.py:174-195
    out = {}
    x = focus(sd, p + ".stem", x)
    out["stem"] = x
    att = lambda i, t: attention(sd, "{}.lsk{}".format(p, i), t) \
        if "{}.lsk{}.proj_1.weight".format(p, i) in sd else t       # new/darknet_att.py:176-201
    for i, name in enumerate(("dark2", "dark3", "dark4")):
        x = any_conv(sd, "{}.{}.0".format(p, name), x, 2)
        x = csp_layer(sd, "{}.{}.1".format(p, name), x, True)
        x = att(i + 2, x)
        out[name] = x
    x = any_conv(sd, p + ".dark5.0", x, 2)
    x = spp_bottleneck(sd, p + ".dark5.1", x)
    x = csp_layer(sd, p + ".dark5.2", x, False)
    x = att(5, x)
    out["dark5"] = x
    return out


def _up2(x: Tensor) -> Tensor:
    return F.interpolate(x, scale_factor=2, mode="nearest")  # nn.Upsample(2,'nearest')


# --------------------------------------------------------------------------- necks
def pafpn(sd: SD, p: str, x: Tensor, with_dark2: bool = False):
    """drone/models/base/yolox.py:170-234 (lsk/yolox6.py:229-293 adds feat0=dark2)."""
    f = csp_darknet(sd, p + ".backbone", x)
    feat1, feat2, feat3 = f["dark3"], f["dark4"], f["dark5"]
    P5 = base_conv(sd, p + ".lateral_conv0", feat3)
    t = csp_layer(sd, p + ".C3_p4", torch.cat((_up2(P5), feat2), 1), False)
    P4 = base_conv(sd, p + ".reduce_conv1", t)
    P3_out = csp_layer(sd, p + ".C3_p3", torch.cat((_up2(P4), feat1), 1), False)
    d = any_conv(sd, p + ".bu_conv2", P3_out, 2)
    P4_out = csp_layer(sd, p + ".C3_n3", torch.cat((d, P4), 1), False)
    d = any_conv(sd, p + ".bu_conv1", P4_out, 2)
    P5_out = csp_layer(sd, p + ".C3_n4", torch.cat((d, P5), 1), False)
    if with_dark2:
        return f["dark2"], P3_out, P4_out, P5_out
    return P3_out, P4_out, P5_out


def non_local_block(sd: SD, p: str, x: Tensor) -> Tensor:
    """x + conv_out( (theta^T phi / N) g^T ) -- dot-product form, divide by N, no softmax.
    drone/models/block/non_local/Identity_Conv.py:152-173."""
    n, _, h, w = x.shape
    g = plain_conv(sd, p + ".g", x).flatten(2).transpose(1, 2)          # [n, N, C]
    th = plain_conv(sd, p + ".theta", x).flatten(2).transpose(1, 2)     # [n, N, C]
    ph = plain_conv(sd, p + ".phi", x).flatten(2)                       # [n, C, N]
    pw = torch.matmul(th, ph)
    pw = pw / pw.shape[-1]
    y = torch.matmul(pw, g).transpose(1, 2).reshape(n, -1, h, w)
    # HIP path: theta|phi|g stored fp16; Gram, fold and conv_out (weights included) in fp32; one store of x + ...
    return plain_conv(sd, p + ".conv_out", y, post=lambda t: x + t, weights_fp32=True)


def _quadrants(x: Tensor):
    # floor split exactly in half whatever patch_scale says: Identity_Conv.py:293-296
    hh, hw = int(x.shape[2] / 2), int(x.shape[3] / 2)
    return x[:, :, :hh, :hw], x[:, :, hh:, :hw], x[:, :, :hh, hw:], x[:, :, hh:, hw:]


def patch_conv(sd: SD, p: str, x: Tensor, stride: int, nonlocal_: bool) -> Tensor:
    """Patch_Conv (Identity_Conv.py:267-320) and Patch_Conv_NonLocal (:323-387):
    per-quadrant 3x3 (stride s) [-> non-local], re-stitch halves l/r/t/b, 3x3 on each,
    stitch lr (along W) and tb (along H), channel-concat, 1x1 'linear' conv with bias."""
    lt, lb, rt, rb = _quadrants(x)
    q = {}
    for name, t in (("lt", lt), ("lb", lb), ("rt", rt), ("rb", rb)):
        t = base_conv(sd, "{}.feat_patchconv_{}".format(p, name), t, stride)
        if nonlocal_:
            t = non_local_block(sd, "{}.feat_patchconv_{}_nonlocal".format(p, name), t)
        q[name] = t
    l = base_conv(sd, p + ".feat_patchconv_l", torch.cat((q["lt"], q["lb"]), 2))
    r = base_conv(sd, p + ".feat_patchconv_r", torch.cat((q["rt"], q["rb"]), 2))
    t = base_conv(sd, p + ".feat_patchconv_t", torch.cat((q["lt"], q["rt"]), 3))
    b = base_conv(sd, p + ".feat_patchconv_b", torch.cat((q["lb"], q["rb"]), 3))
    both = torch.cat((torch.cat((l, r), 3), torch.cat((t, b), 2)), 1)
    if p + ".channel_conv.weight" in sd:           # channel_cat == 'linear'
        return plain_conv(sd, p + ".channel_conv", both)
    return base_conv(sd, p + ".channel_conv", both)


def patch_conv_nonlocal_new(sd: SD, p: str, x: Tensor) -> Tensor:
    """Patch_Conv_NonLocal_new (drone/models/new/Non_local_family.py:208-252): a non-local block
    on each floor-split quadrant AT the input resolution, re-stitch, then channel_conv
    (BaseConv 3x3 for channel_cat='non_linear', Conv2d 1x1+bias for 'linear')."""
    lt, lb, rt, rb = _quadrants(x)
    q = {name: non_local_block(sd, "{}.feat_patchconv_{}_nonlocal".format(p, name), t)
         for name, t in (("lt", lt), ("lb", lb), ("rt", rt), ("rb", rb))}
    top = torch.cat((q["lt"], q["rt"]), 3)
    bot = torch.cat((q["lb"], q["rb"]), 3)
    both = torch.cat((top, bot), 2)
    if p + ".channel_conv.weight" in sd:
        return plain_conv(sd, p + ".channel_conv", both)
    return base_conv(sd, p + ".channel_conv", both)


def patch_conv_nonlocal_44(sd: SD, p: str, x: Tensor) -> Tensor:
    """Patch_Conv_NonLocal_44 (drone/models/new/Non_local_family.py:359-421): a Patch_Conv_NonLocal (stride 2, with its own
    2x2 split) on each quadrant, the four results re-stitched into halves l/r/t/b, a 1x1 BaseConv on each, lr (along W)
    and tb (along H) concatenated on channels, channel_conv."""
    lt, lb, rt, rb = _quadrants(x)
    q = {name: patch_conv(sd, "{}.patchconv_{}_nonlocal".format(p, name), t, 2, True)
         for name, t in (("lt", lt), ("lb", lb), ("rt", rt), ("rb", rb))}
    l = base_conv(sd, p + ".feat_patchconv_l", torch.cat((q["lt"], q["lb"]), 2))
    r = base_conv(sd, p + ".feat_patchconv_r", torch.cat((q["rt"], q["rb"]), 2))
    t = base_conv(sd, p + ".feat_patchconv_t", torch.cat((q["lt"], q["rt"]), 3))
    b = base_conv(sd, p + ".feat_patchconv_b", torch.cat((q["lb"], q["rb"]), 3))
    both = torch.cat((torch.cat((l, r), 3), torch.cat((t, b), 2)), 1)
    if p + ".channel_conv.weight" in sd:
        return plain_conv(sd, p + ".channel_conv", both)
    return base_conv(sd, p + ".channel_conv", both)


def get_centroid(x: Tensor) -> Tuple[int, int]:
    """Patch_Conv_NonLocal_adapt_new.get_centroid (drone/models/new/Non_local_family.py:298-321): the first column /
    row at which the running sum (over the WHOLE batch) exceeds half of the total, rounded down to even and clamped to
    [4, size - 4].  -> (centroid_x: split along H, centroid_y: split along W)."""
    x_2, x_3 = x.sum(2), x.sum(3)
    total = x.sum()

    def first(v, n):
        d = 0
        i = 0
        for i in range(n):
            d = v[:, :, i] + d
            if d.sum() > 0.5 * total:
                break
        i = i // 2 * 2
        i = 4 if i < 4 else i
        return n - 4 if i > n - 4 else i
    return first(x_3, x.shape[2]), first(x_2, x.shape[3])


def adapt_split(att: Tensor) -> Tuple[int, int, int]:
    """The data-dependent quadrant split of Patch_Conv_NonLocal_adapt_new.forward (:324-334) from the attention map:
    values under min + 0.75 (max - min) are zeroed, then (row split, column split of the top part, of the bottom part)."""
    a = att.clone()
    mx, mn = a.max(), a.min()
    a[a < mn + 0.75 * (mx - mn)] = 0
    cx, _ = get_centroid(a)
    _, cyl = get_centroid(a[:, :, :cx, :])
    _, cyr = get_centroid(a[:, :, cx:, :])
    return cx, cyl, cyr


def patch_conv_nonlocal_adapt_new(sd: SD, p: str, x: Tensor) -> Tensor:
    """Patch_Conv_NonLocal_adapt_new (Non_local_family.py:272-357): the quadrant split follows the thresholded spatial
    attention map (one split for the whole batch, App. D.3), non-local per quadrant at the input resolution, a 3x3
    BaseConv on the top and on the bottom part, channel_conv, and the result gated by the (unthresholded) attention map."""
    att = spatial_attention(sd, p + ".attention_map", x)
    cx, cyl, cyr = adapt_split(att)
    nl = lambda name, t: non_local_block(sd, "{}.feat_patchconv_{}_nonlocal".format(p, name), t)
    lt, lb = nl("lt", x[:, :, :cx, :cyl]), nl("lb", x[:, :, cx:, :cyr])
    rt, rb = nl("rt", x[:, :, :cx, cyl:]), nl("rb", x[:, :, cx:, cyr:])
    t = base_conv(sd, p + ".feat_patchconv_t", torch.cat((lt, rt), 3))
    b = base_conv(sd, p + ".feat_patchconv_b", torch.cat((lb, rb), 3))
    both = torch.cat((t, b), 2)
    if p + ".channel_conv.weight" in sd:
        y = plain_conv(sd, p + ".channel_conv", both)
    else:
        y = base_conv(sd, p + ".channel_conv", both)
    return _q(spatial_attention(sd, p + ".attention_map", x) * y, p + ".gated")


def patch_conv_nonlocal_adapt(sd: SD, p: str, x: Tensor) -> Tensor:
    """Patch_Conv_NonLocal_adapt (Non_local_family.py:112-206): the adaptive split as above, then per quadrant a stride-2
    3x3 BaseConv and a non-local block, a 3x3 BaseConv on the re-joined top and bottom parts, channel_conv.  (No gating.)"""
    att = spatial_attention(sd, p + ".attention_map", x)
    cx, cyl, cyr = adapt_split(att)
    q = {}
    for name, t in (("lt", x[:, :, :cx, :cyl]), ("lb", x[:, :, cx:, :cyr]), ("rt", x[:, :, :cx, cyl:]), ("rb", x[:, :, cx:, cyr:])):
        t = base_conv(sd, "{}.feat_patchconv_{}".format(p, name), t, 2)
        q[name] = non_local_block(sd, "{}.feat_patchconv_{}_nonlocal".format(p, name), t)
    t = base_conv(sd, p + ".feat_patchconv_t", torch.cat((q["lt"], q["rt"]), 3))
    b = base_conv(sd, p + ".feat_patchconv_b", torch.cat((q["lb"], q["rb"]), 3))
    both = torch.cat((t, b), 2)
    if p + ".channel_conv.weight" in sd:
        return plain_conv(sd, p + ".channel_conv", both)
    return base_conv(sd, p + ".channel_conv", both)


def lsk_block(sd: SD, p: str, x: Tensor) -> Tensor:
    """LSKblock (drone/models/lsk/LSK.py:27-51): depthwise 5x5 -> depthwise 7x7 dilation 3 -> a 1x1 each to dim/2 ->
    [mean_c, max_c] of their concat -> 7x7 conv 2->2 -> sigmoid -> attn1 * sig0 + attn2 * sig1 -> 1x1 to dim -> x * attn.
    (The HIP path stores every intermediate; the fp16-storage emulation rounds at the same points.)"""
    dim = x.shape[1]
    a1 = _q(F.conv2d(x, _qw(sd[p + ".conv0.weight"], p + ".conv0"), sd[p + ".conv0.bias"], 1, 2, 1, dim), p + ".conv0")
    a2 = _q(F.conv2d(a1, _qw(sd[p + ".conv_spatial.weight"], p + ".conv_spatial"), sd[p + ".conv_spatial.bias"], 1, 9, 3, dim),
            p + ".conv_spatial")
    a1 = plain_conv(sd, p + ".conv1", a1)
    a2 = plain_conv(sd, p + ".conv2", a2)
    attn = torch.cat((a1, a2), 1)
    agg = _q(torch.cat((attn.mean(1, keepdim=True), attn.max(1, keepdim=True)[0]), 1), p + ".agg")
    sig = plain_conv(sd, p + ".conv_squeeze", agg, 1, 3, post=torch.sigmoid)
    mix = _q(a1 * sig[:, 0:1] + a2 * sig[:, 1:2], p + ".mix")
    attn = plain_conv(sd, p + ".conv", mix)
    return _q(x * attn, p + ".out")


def attention(sd: SD, p: str, x: Tensor) -> Tensor:
    """Attention (Non_local_family.py:254-272; lsk/LSK.py:54-71 with an LSKblock): proj_1 1x1 -> exact GELU -> gating unit
    (told from the parameter names: the quadrant non-local unit, its attention-split form, or the LSK block) -> proj_2
    1x1 -> + shortcut."""
    y = plain_conv(sd, p + ".proj_1", x, post=F.gelu)
    g = p + ".spatial_gating_unit"
    if g + ".conv_spatial.weight" in sd:
        y = lsk_block(sd, g, y)
    elif g + ".attention_map.conv.weight" in sd:
        y = patch_conv_nonlocal_adapt_new(sd, g, y)
    else:
        y = patch_conv_nonlocal_new(sd, g, y)
    return plain_conv(sd, p + ".proj_2", y, post=lambda t: t + x)


def spatial_attention(sd: SD, p: str, x: Tensor) -> Tensor:
    """SpatialAttention (Non_local_family.py:423-436): sigmoid(conv7x7([max_c x, mean_c x]))."""
    k = sd[p + ".conv.weight"].shape[-1]
    r = _q(torch.cat((x.max(1, keepdim=True)[0], x.mean(1, keepdim=True)), 1), p + ".maxmean")
    return plain_conv(sd, p + ".conv", r, 1, k // 2, post=torch.sigmoid)


def csp_darknet_att(sd: SD, p: str, x: Tensor) -> Dict[str, Tensor]:
    """drone/models/new/darknet_att.py:120-203: CSPDarknet with an Attention block (lsk2..lsk5)
    after every stage.  csp_darknet() below applies them when the keys are present."""
    return csp_darknet(sd, p, x)


def identity_conv(sd: SD, p: str, x: Tensor) -> Tensor:
    # Identity_Conv_{three,five,seven}: dense kxk conv with bias, pad k//2
    # (Identity_Conv.py:27-84); identity only at init, trained weights are arbitrary.
    k = sd[p + ".conv.weight"].shape[-1]
    return plain_conv(sd, p + ".conv", x, 1, k // 2)


def gl_pafpn(sd: SD, p: str, x: Tensor):
    """GL-fusion neck: drone/models/block/non_local/yolo_patch_nonlocal_plus.py:180-247."""
    f = csp_darknet(sd, p + ".backbone", x)
    feat1, feat2, feat3 = f["dark3"], f["dark4"], f["dark5"]
    feat1_patch = patch_conv(sd, p + ".Patch_conv_feat1", feat1, 2, True)    # global branch
    feat2_patch = patch_conv(sd, p + ".Patch_conv_feat2", feat2, 1, False)   # local branch
    P5 = base_conv(sd, p + ".lateral_conv0", feat3)
    t = csp_layer(sd, p + ".C3_p4", torch.cat((_up2(P5), feat2, feat1_patch), 1), False)
    P4 = base_conv(sd, p + ".reduce_conv1", t)
    P3_out = csp_layer(sd, p + ".C3_p3", torch.cat((_up2(P4), feat1), 1), False)
    P3_out = identity_conv(sd, p + ".P3_Identity", P3_out)
    d = any_conv(sd, p + ".bu_conv2", P3_out, 2)
    P4_out = csp_layer(sd, p + ".C3_n3", torch.cat((d, P4, feat2_patch), 1), False)
    P4_out = identity_conv(sd, p + ".P4_Identity", P4_out)
    d = any_conv(sd, p + ".bu_conv1", P4_out, 2)
    P5_out = csp_layer(sd, p + ".C3_n4", torch.cat((d, P5), 1), False)
    P5_out = identity_conv(sd, p + ".P5_Identity", P5_out)
    return P3_out, P4_out, P5_out


# --------------------------------------------------------------------------- heads
def _tower(sd: SD, p: str, x: Tensor) -> Tensor:
    i = 0
    while ("{}.{}.conv.weight".format(p, i) in sd) or ("{}.{}.dconv.conv.weight".format(p, i) in sd):
        x = any_conv(sd, "{}.{}".format(p, i), x)
        i += 1
    return x


def yolox_head(sd: SD, p: str, feats: Sequence[Tensor]) -> List[Tensor]:
    # drone/models/base/yolox.py:46-92 ; output channel order reg(4), obj(1), cls(nc)
    outs = []
    for k, x in enumerate(feats):
        x = base_conv(sd, "{}.stems.{}".format(p, k), x)
        cls_feat = _tower(sd, "{}.cls_convs.{}".format(p, k), x)
        reg_feat = _tower(sd, "{}.reg_convs.{}".format(p, k), x)
        outs.append(torch.cat((plain_conv(sd, "{}.reg_preds.{}".format(p, k), reg_feat, store=False),
                               plain_conv(sd, "{}.obj_preds.{}".format(p, k), reg_feat, store=False),
                               plain_conv(sd, "{}.cls_preds.{}".format(p, k), cls_feat, store=False)), 1))
    return outs


def cross_scale_head(sd: SD, p: str, feats: Sequence[Tensor]) -> List[Tensor]:
    """Cross-scale decoupled head, drone/models/lsk/yolox6.py:69-153
    (text-identical to new/yolox6.py).  feats = (dark2, P3, P4, P5)."""
    feat0 = csp_layer(sd, p + ".csp_feat0", feats[0], False)
    lv = [base_conv(sd, "{}.stems.{}".format(p, k), x) for k, x in enumerate(feats[1:])]

    def down(k, t):      # up_convs[k]: 3x3 s1 then 3x3 s2
        t = any_conv(sd, "{}.up_convs.{}.0".format(p, k), t, 1)
        return any_conv(sd, "{}.up_convs.{}.1".format(p, k), t, 2)

    outs = []
    for k, x in enumerate(lv):
        finer = feat0 if k == 0 else lv[k - 1]
        parts = [x, down(k, finer)]
        if k < len(lv) - 1:
            parts.append(_up2(lv[k + 1]))
        cls_feat = _tower(sd, "{}.cls_convs.{}".format(p, k), torch.cat(parts, 1))
        reg_feat = _tower(sd, "{}.reg_convs.{}".format(p, k), x)
        outs.append(torch.cat((plain_conv(sd, "{}.reg_preds.{}".format(p, k), reg_feat, store=False),
                               plain_conv(sd, "{}.obj_preds.{}".format(p, k), reg_feat, store=False),
                               plain_conv(sd, "{}.cls_preds.{}".format(p, k), cls_feat, store=False)), 1))
    return outs


# --------------------------------------------------------------------------- whole models
def yolox_base_forward(sd: SD, x: Tensor) -> List[Tensor]:
    """models.base.yolox.YoloBody.forward  (drone/models/base/yolox.py:248-251)."""
    return yolox_head(sd, "head", pafpn(sd, "backbone", x))


def yolox_gl_forward(sd: SD, x: Tensor) -> List[Tensor]:
    """models.block.non_local.yolo_patch_nonlocal_plus.YoloBody.forward (:260-263)."""
    return yolox_head(sd, "head", gl_pafpn(sd, "backbone", x))


def yolox_cross_forward(sd: SD, x: Tensor) -> List[Tensor]:
    """models.lsk.yolox6.YoloBody.forward (drone/models/lsk/yolox6.py:309-312).
    NOTE: that file's backbone is darknet_lsk.CSPDarknet; this restatement covers the
    plain-CSPDarknet twin (new/yolox6.py, whose `models.decouple` import is missing)."""
    return cross_scale_head(sd, "head", pafpn(sd, "backbone", x, with_dark2=True))


FORWARDS = {"base": yolox_base_forward, "gl": yolox_gl_forward, "cross": yolox_cross_forward}


# --------------------------------------------------------------------------- post-process
def decode_outputs(outputs: Sequence[Tensor], input_shape: Sequence[int]) -> Tensor:
    """drone/models/core/utils_bbox.py:254-306.  [B,5+nc,H,W]x3 -> [B,A,5+nc] with
    sigmoid on [4:], xy=(xy+grid)*stride, wh=exp(wh)*stride, normalised by (W,H).
    stride = input_shape[0] / h for both axes (Appendix D.6).  Does not mutate inputs."""
    flat = torch.cat([o.flatten(2) for o in outputs], 2).permute(0, 2, 1).clone()
    flat[:, :, 4:] = torch.sigmoid(flat[:, :, 4:])
    grids, strides = [], []
    for o in outputs:
        h, w = o.shape[-2:]
        gy, gx = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
        grids.append(torch.stack((gx, gy), 2).reshape(1, -1, 2).to(flat.dtype))
        strides.append(torch.full((1, h * w, 1), input_shape[0] / h, dtype=flat.dtype))
    grids, strides = torch.cat(grids, 1), torch.cat(strides, 1)
    flat[..., :2] = (flat[..., :2] + grids) * strides
    flat[..., 2:4] = torch.exp(flat[..., 2:4]) * strides
    flat[..., [0, 2]] = flat[..., [0, 2]] / input_shape[1]
    flat[..., [1, 3]] = flat[..., [1, 3]] / input_shape[0]
    return flat


def nms_single(boxes: np.ndarray, scores: np.ndarray, thr: float) -> np.ndarray:
    """Greedy NMS, torchvision.ops.nms semantics (source not in /root/reference; restated
    from its documented behaviour): visit by descending score,
    suppress when IoU > thr, areas without '+1', result ordered by descending score."""
    order = np.argsort(-scores, kind="stable")
    x1, y1, x2, y2 = (boxes[:, i].astype(np.float32) for i in range(4))
    areas = (x2 - x1) * (y2 - y1)
    dead = np.zeros(len(order), bool)
    keep = []
    for a in range(len(order)):
        i = order[a]
        if dead[i]:
            continue
        keep.append(i)
        rest = order[a + 1:]
        w = np.maximum(np.float32(0), np.minimum(x2[i], x2[rest]) - np.maximum(x1[i], x1[rest]))
        h = np.maximum(np.float32(0), np.minimum(y2[i], y2[rest]) - np.maximum(y1[i], y1[rest]))
        inter = w * h
        with np.errstate(divide="ignore", invalid="ignore"):
            iou = inter / (areas[i] + areas[rest] - inter)
        dead[rest[iou > np.float32(thr)]] = True
    return np.asarray(keep, np.int64)


def batched_nms(boxes: np.ndarray, scores: np.ndarray, labels: np.ndarray, thr: float) -> np.ndarray:
    """torchvision.ops.boxes.batched_nms semantics (per-class NMS; result indices sorted
    by descending score) in its 'vanilla' per-class-loop form, which torchvision
    documents as equivalent to the coordinate-offset form.  PARITY UNPINNED."""
    keep = []
    for c in np.unique(labels):
        idx = np.nonzero(labels == c)[0]
        keep.append(idx[nms_single(boxes[idx], scores[idx], thr)])
    if not keep:
        return np.zeros((0,), np.int64)
    keep = np.concatenate(keep)
    return keep[np.argsort(-scores[keep], kind="stable")]


def yolo_correct_boxes(box_xy, box_wh, input_shape, image_shape, letterbox_image):
    # drone/models/core/utils_bbox.py:8-33 ; returns [y1,x1,y2,x2] in original-image pixels
    box_yx, box_hw = box_xy[..., ::-1], box_wh[..., ::-1]
    input_shape = np.array(input_shape, np.float64)
    image_shape = np.array(image_shape, np.float64)
    if letterbox_image:
        new_shape = np.round(image_shape * np.min(input_shape / image_shape))
        offset = (input_shape - new_shape) / 2.0 / input_shape
        scale = input_shape / new_shape
        box_yx = (box_yx - offset) * scale
        box_hw = box_hw * scale
    mins, maxes = box_yx - box_hw / 2.0, box_yx + box_hw / 2.0
    out = np.concatenate([mins[..., 0:1], mins[..., 1:2], maxes[..., 0:1], maxes[..., 1:2]], -1)
    return out * np.concatenate([image_shape, image_shape], -1)


def non_max_suppression(prediction: Tensor, num_classes: int, input_shape, image_shape,
                        letterbox_image: bool, conf_thres: float = 0.5, nms_thres: float = 0.4):
    """drone/models/core/utils_bbox.py:375-484.  prediction [B,A,5+nc] (decode_outputs
    result).  Per image: None, or ndarray(n,7) = [y1,x1,y2,x2,obj,cls_conf,cls_id]."""
    pred = prediction.clone().float()
    cxcywh = pred[:, :, :4].clone()
    pred[:, :, 0] = cxcywh[:, :, 0] - cxcywh[:, :, 2] / 2
    pred[:, :, 1] = cxcywh[:, :, 1] - cxcywh[:, :, 3] / 2
    pred[:, :, 2] = cxcywh[:, :, 0] + cxcywh[:, :, 2] / 2
    pred[:, :, 3] = cxcywh[:, :, 1] + cxcywh[:, :, 3] / 2
    out: List[Optional[np.ndarray]] = [None] * len(pred)
    for i, ip in enumerate(pred):
        if not ip.size(0):
            continue
        class_conf, class_pred = torch.max(ip[:, 5:5 + num_classes], 1, keepdim=True)
        mask = ip[:, 4] * class_conf[:, 0] >= conf_thres
        det = torch.cat((ip[:, :5], class_conf, class_pred.float()), 1)[mask].numpy()
        keep = batched_nms(det[:, :4], det[:, 4] * det[:, 5], det[:, 6], nms_thres)
        det = det[keep]
        xy, wh = (det[:, 0:2] + det[:, 2:4]) / 2, det[:, 2:4] - det[:, 0:2]
        det[:, :4] = yolo_correct_boxes(xy, wh, input_shape, image_shape, letterbox_image)
        out[i] = det
    return out


# --------------------------------------------------------------------------- synthetic weights
# The deterministic (key, shape, seed) -> tensor filler is input-data generation, not
# arithmetic of the path; it lives in glsdet_amd/synth.py so bench.py can build weights
# without importing the oracle.  Re-exported here for the tests and the golden generator.
from glsdet_amd.synth import synth_input, synth_state_dict, synth_tensor  # noqa: E402,F401


# --------------------------------------------------------------------------- mmdet flavour
def mlvl_point_priors(sizes, strides) -> Tensor:
    """MlvlPointGenerator(strides, offset=0).grid_priors(sizes, with_stride=True), levels concatenated: [x, y, s, s] per
    position, x fastest (ufp/mmdet/core/anchor/point_generator.py:100-177).  Pinned by tests/golden/yolox_mmdet_golden.npz."""
    out = []
    for (h, w), s in zip(sizes, strides):
        gy, gx = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
        out.append(torch.stack((gx.flatten() * s, gy.flatten() * s, torch.full((h * w,), s), torch.full((h * w,), s)), 1).float())
    return torch.cat(out)


def mmdet_bbox_decode(priors: Tensor, preds: Tensor) -> Tensor:
    """YOLOXHead._bbox_decode (ufp/mmdet/models/dense_heads/yolox_head.py:298-308): centre = pred_xy * stride + prior_xy,
    size = exp(pred_wh) * stride, -> x1, y1, x2, y2."""
    xys = preds[..., :2] * priors[:, 2:] + priors[:, :2]
    whs = preds[..., 2:4].exp() * priors[:, 2:]
    return torch.cat((xys - whs / 2, xys + whs / 2), -1)


def mmdet_yolox_get_bboxes(outs: Sequence[Tensor], num_classes: int, strides: Sequence[int], score_thr: float,
                           iou_thr: float, scale_factors=None):
    """YOLOXHead.get_bboxes / _bbox_decode / _bboxes_nms + bbox2result restated
    (ufp/mmdet/models/dense_heads/yolox_head.py:249-322, core/anchor/point_generator.py with
    offset=0, core/bbox/transforms.py:116-133).  `outs` = per level [B, 4+1+nc, H, W] in the
    drone channel order (reg, obj, cls).  -> list[img] of list[class] of ndarray(n,5)."""
    B = outs[0].shape[0]
    flat = [o.permute(0, 2, 3, 1).reshape(B, -1, o.shape[1]) for o in outs]
    priors, flat = mlvl_point_priors([o.shape[-2:] for o in outs], strides), torch.cat(flat, 1).float()
    boxes = mmdet_bbox_decode(priors, flat[..., :4])
    if scale_factors is not None:
        boxes = boxes / torch.as_tensor(np.asarray(scale_factors, np.float32)).reshape(B, 1, 4)
    obj, cls = flat[..., 4].sigmoid(), flat[..., 5:5 + num_classes].sigmoid()
    results = []
    for i in range(B):
        max_scores, labels = torch.max(cls[i], 1)
        valid = obj[i] * max_scores >= score_thr
        b, s, l = boxes[i][valid].numpy(), (max_scores[valid] * obj[i][valid]).numpy(), labels[valid].numpy()
        if len(l) == 0:
            results.append([np.zeros((0, 5), np.float32) for _ in range(num_classes)])
            continue
        keep = batched_nms(b, s, l.astype(np.float32), iou_thr)
        dets = np.concatenate([b[keep], s[keep][:, None]], 1).astype(np.float32)
        lk = l[keep]
        results.append([dets[lk == c] for c in range(num_classes)])
    return results
