#!/usr/bin/env python3
"""Golden-vector generator -- runs ONLY in the build container, where the reference is
mounted read-only at /root/reference.  It imports the reference's own CPU implementation
(yolox-drone tree), fills every parameter/buffer with the deterministic filler
``oracle.glsdet_oracle.synth_tensor`` (so no state_dict has to be stored), runs the
reference forward and writes inputs' seeds + expected outputs to ``tests/golden/*.npz``
and the reference's state_dict key->shape tables to ``tests/golden/shapes_*.json``.

The reference never travels to the GPU box; these small data files do.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/yolox-drone")
sys.dont_write_bytecode = True

from oracle.glsdet_oracle import synth_input, synth_tensor  # noqa: E402

torch.set_grad_enabled(False)
torch.manual_seed(0)


def fill(module, seed):
    sd = module.state_dict()
    new = {k: torch.from_numpy(synth_tensor(k, tuple(v.shape), seed)) for k, v in sd.items()}
    module.load_state_dict(new)
    module.eval()
    return {k: list(v.shape) for k, v in sd.items()}


def calibrate_bn(net, x, seed):
    """Whole models only: replace the filler's BN running stats by perturbed batch
    statistics of a synthetic input, so the signal neither dies nor explodes over ~80
    layers and the logits stay input-sensitive.  Done in ONE eval-mode forward with a
    pre-hook per BN (execution order = topological order), so every BN is calibrated on
    what its already-perturbed upstream really produces.
    Returns {key: ndarray} for every running_mean / running_var (stored in the fixture:
    these are the only tensors that are NOT a pure function of (key, shape, seed))."""
    rng = np.random.default_rng([seed, 0xB17])

    def pre(m, inp):
        t = inp[0]
        var = t.var((0, 2, 3), unbiased=False)
        var = var + 0.5 * var.mean() + 1e-4      # floor: no channel may amplify by > ~1.4x
        mean = t.mean((0, 2, 3))
        c = mean.numel()
        m.running_var.copy_(var * torch.from_numpy(rng.uniform(0.8, 1.25, c).astype(np.float32)))
        m.running_mean.copy_(mean + var.sqrt() * torch.from_numpy(
            (0.1 * rng.standard_normal(c)).astype(np.float32)))

    hooks = [m.register_forward_pre_hook(pre) for m in net.modules()
             if isinstance(m, torch.nn.BatchNorm2d)]
    net.eval()
    net(x)
    for h in hooks:
        h.remove()
    return {k: v.numpy().copy() for k, v in net.state_dict().items()
            if k.endswith("running_mean") or k.endswith("running_var")}


def attention_cases(block):
    """SURVEY section 8a row A6': the other GL attention variants the tree holds
    (drone/models/new/Non_local_family.py, new/darknet_att.py)."""
    from models.new.Non_local_family import Attention, Patch_Conv_NonLocal_new, SpatialAttention
    from models.new.darknet_att import CSPDarknet as AttDarknet
    block("att_pcnl_new_nonlinear", lambda: Patch_Conv_NonLocal_new(32, 32, channel_scale=1), (2, 32, 20, 24))
    block("att_pcnl_new_linear", lambda: Patch_Conv_NonLocal_new(32, 48, channel_scale=1, channel_cat="linear"),
          (2, 32, 21, 27))
    block("att_attention_c32", lambda: Attention(32), (2, 32, 20, 24))
    block("att_attention_c48_odd", lambda: Attention(48), (1, 48, 17, 23))
    block("att_spatial_attention", lambda: SpatialAttention(7), (2, 32, 20, 24))

    class Outs(torch.nn.Module):            # dict -> tuple so that block() can store it
        def __init__(self):
            super().__init__()
            self.backbone = AttDarknet(0.33, 0.375, out_features=("dark2", "dark3", "dark4", "dark5"))

        def forward(self, x):
            f = self.backbone(x)
            return torch.cat([f[k].flatten(1) for k in ("dark2", "dark3", "dark4", "dark5")], 1)
    block("att_darknet_tiny", Outs, (1, 3, 128, 160), calibrate=True)


PRE_CASES = [  # (in_h, in_w), input_shape (H, W), letterbox   -- synth_image(seed = index)
    ((77, 123), (64, 96), False), ((77, 123), (64, 96), True), ((150, 90), (64, 96), True),
    ((64, 96), (64, 96), False), ((300, 500), (64, 96), False), ((31, 45), (64, 96), True),
    ((97, 64), (96, 64), True),
]


def synth_image(shape, seed):
    """Smooth colour gradients + texture + saturated patches, uint8 HWC RGB (test input data)."""
    h, w = shape
    rng = np.random.default_rng([seed, 0x1A6E])
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    img = np.stack([127 + 120 * np.sin(xx / (5 + 3 * c) + seed) * np.cos(yy / (7 + 2 * c)) for c in range(3)], -1)
    img += rng.normal(0, 25, img.shape)
    img[: h // 5, : w // 4] = 255
    img[-(h // 6):, -(w // 5):] = 0
    return np.clip(img, 0, 255).astype(np.uint8)


def preprocess_cases():
    """Row f.3: the reference's own resize_image + preprocess_input (drone/models/core/utils.py)."""
    from PIL import Image
    from models.core.utils import preprocess_input, resize_image
    out = {}
    for i, (ishape, shape, lb) in enumerate(PRE_CASES):
        img = Image.fromarray(synth_image(ishape, i), "RGB")
        data = resize_image(img, (shape[1], shape[0]), lb)
        out["pre/%d" % i] = np.expand_dims(np.transpose(preprocess_input(np.array(data, dtype="float32")), (2, 0, 1)), 0)
    np.savez_compressed(os.path.join(HERE, "preprocess_golden.npz"), **out)
    print("preprocess:", len(out), "cases,", os.path.getsize(os.path.join(HERE, "preprocess_golden.npz")), "bytes")


def ufp_boxes(trial):
    """Seeded coarse-detection-like xyxy boxes for one image (input data of the packing goldens)."""
    rng = np.random.default_rng([trial, 0x0F9])
    n = int(rng.integers(1, 60))
    W, H = int(rng.integers(400, 2000)), int(rng.integers(300, 1200))
    c = rng.uniform(0, 1, (n, 2)) * [W, H]
    wh = np.exp(rng.uniform(np.log(6), np.log(220), (n, 2)))
    b = np.concatenate([c - wh / 2, c + wh / 2], 1)
    b[:, 0::2] = np.clip(b[:, 0::2], 0, W - 1)
    b[:, 1::2] = np.clip(b[:, 1::2], 0, H - 1)
    return (b.astype(np.float32) if trial % 3 == 0 else b), W, H


def ufp_cases():
    """Rows f.1: the reference's UnifiedForegroundPacking (ufp/UFPMP-Det-Tools/ufp, pure numpy)."""
    sys.path.insert(0, "/root/reference/yolox-ufp/UFPMP-Det-Tools")
    from ufp import UnifiedForegroundPacking
    out = {}
    for trial in range(24):
        b, W, H = ufp_boxes(trial)
        chips, cw, ch = UnifiedForegroundPacking(b.copy(), 1.5, input_shape=[W, H])
        out["ufp/%d/chips" % trial] = np.asarray(chips, np.float64).reshape(-1, 7)
        out["ufp/%d/canvas" % trial] = np.asarray([cw, ch], np.float64)
    np.savez_compressed(os.path.join(HERE, "ufp_golden.npz"), **out)
    print("ufp:", len(out) // 2, "cases,", os.path.getsize(os.path.join(HERE, "ufp_golden.npz")), "bytes")


def main():
    ufp_cases()
    preprocess_cases()
    from models.base import yolox as ref_base
    from models.base.baseConv import BaseConv, DWConv
    from models.base.darknet import Bottleneck, CSPLayer, Focus, SPPBottleneck
    from models.block.non_local import yolo_patch_nonlocal_plus as ref_gl
    from models.block.non_local.Identity_Conv import (Identity_Conv_five, Identity_Conv_seven,
                                                      Identity_Conv_three, Non_local_Block,
                                                      Patch_Conv, Patch_Conv_NonLocal)
    # decode_outputs is pure torch but its module imports torchvision (absent) at the top:
    # register an empty stand-in so the import succeeds; NMS itself is never called here.
    if "torchvision" not in sys.modules:
        tv = types.ModuleType("torchvision")
        ops = types.ModuleType("torchvision.ops")
        ops.boxes = types.ModuleType("torchvision.ops.boxes")
        tv.ops = ops
        sys.modules.update({"torchvision": tv, "torchvision.ops": ops,
                            "torchvision.ops.boxes": ops.boxes})
    from models.core.utils_bbox import decode_outputs, yolo_correct_boxes

    out = {}

    # ------------------------------------------------------------------ blocks
    def block(tag, ctor, in_shape, seed=0, calibrate=False):
        m = ctor()
        shapes = fill(m, seed)
        x = synth_input(in_shape, seed + 100)
        if calibrate:
            for k, v in calibrate_bn(m, x, seed).items():
                out["block/%s/bn/%s" % (tag, k)] = v
        y = m(x)
        out["block/%s/y" % tag] = y.numpy()
        out["block/%s/meta" % tag] = np.frombuffer(json.dumps(
            {"shapes": shapes, "in_shape": list(in_shape), "seed": seed}).encode(), np.uint8)

    for k in (1, 3):
        for s in (1, 2):
            for act in ("silu", "relu", "lrelu"):
                block("baseconv_k%d_s%d_%s" % (k, s, act),
                      lambda k=k, s=s, act=act: BaseConv(16, 32, k, s, act=act), (2, 16, 20, 24))
    block("dwconv_k3_s2", lambda: DWConv(16, 32, 3, 2), (2, 16, 20, 24))
    block("focus", lambda: Focus(3, 16, 3), (2, 3, 32, 40))
    block("spp", lambda: SPPBottleneck(32, 32), (2, 32, 20, 24))
    block("bottleneck_add", lambda: Bottleneck(16, 16, True, 1.0), (2, 16, 20, 24))
    block("bottleneck_noadd", lambda: Bottleneck(16, 16, False, 1.0), (2, 16, 20, 24))
    block("csp_n2_shortcut", lambda: CSPLayer(32, 32, 2, True), (2, 32, 20, 24))
    block("csp_n1_noshortcut", lambda: CSPLayer(48, 32, 1, False), (2, 48, 20, 24))
    block("nonlocal_c16", lambda: Non_local_Block(16, 16), (2, 16, 10, 12))
    block("nonlocal_c32_inter16", lambda: Non_local_Block(32, None), (2, 32, 5, 7))
    block("patch_conv_s1", lambda: Patch_Conv(32, 16, patch_scale=4, stride=1), (2, 32, 20, 24))
    block("patch_conv_nonlocal_s2", lambda: Patch_Conv_NonLocal(16, 32, patch_scale=2), (2, 16, 40, 48))
    block("identity3", lambda: Identity_Conv_three(16, 16), (2, 16, 20, 24))
    block("identity5", lambda: Identity_Conv_five(16, 16), (2, 16, 20, 24))
    block("identity7", lambda: Identity_Conv_seven(16, 16), (2, 16, 20, 24))

    # ------------------------------------------------------------------ cross-scale decoupled head
    # drone/models/lsk/yolox6.py:7-153 (text-identical to new/yolox6.py): the head class alone,
    # fed with four synthetic pyramid features (dark2, P3, P4, P5 of a width-0.5 model).
    from models.lsk.yolox6 import YOLOXHead as CrossHead
    for seed in (0, 1):
        m = CrossHead(10, 0.5)
        shapes = fill(m, seed)
        feats = [synth_input(sh, seed + 300 + i) for i, sh in enumerate(
            [(2, 64, 32, 40), (2, 128, 16, 20), (2, 256, 8, 10), (2, 512, 4, 5)])]
        ys = m(feats)
        for i, y in enumerate(ys):
            out["crosshead/seed%d/out%d" % (seed, i)] = y.numpy()
        out["crosshead/seed%d/meta" % seed] = np.frombuffer(json.dumps(
            {"shapes": shapes, "seed": seed, "feat_shapes": [list(f.shape) for f in feats]}).encode(), np.uint8)

    # ------------------------------------------------------------------ whole models
    shapes_all = {}
    for mname, mod in (("base", ref_base), ("gl", ref_gl)):
        for phi in ("nano", "tiny", "s"):
            for seed in (0, 1):
                if phi == "s" and seed == 1:
                    continue
                net = mod.YoloBody(10, phi)
                shapes = fill(net, seed)
                shapes_all["%s_%s" % (mname, phi)] = shapes
                in_shape = (2, 3, 128, 160) if phi != "s" else (1, 3, 128, 160)
                x = synth_input(in_shape, seed + 100)
                tag = "model/%s_%s_seed%d" % (mname, phi, seed)
                for k, v in calibrate_bn(net, x, seed).items():
                    out["%s/bn/%s" % (tag, k)] = v
                ys = net(x)
                for i, y in enumerate(ys):
                    out["%s/out%d" % (tag, i)] = y.numpy()
                dec = decode_outputs([y.clone() for y in ys], list(in_shape[2:]))
                out["%s/decoded" % tag] = dec.numpy()
                out["%s/meta" % tag] = np.frombuffer(json.dumps(
                    {"in_shape": list(in_shape), "seed": seed, "num_classes": 10, "phi": phi,
                     "model": mname}).encode(), np.uint8)
                print(tag, [tuple(y.shape) for y in ys],
                      "logit std %.3f" % float(torch.cat([y.flatten() for y in ys]).std()))

    # ------------------------------------------------------------------ yolo_correct_boxes
    rng = np.random.default_rng(7)
    xy = rng.uniform(0.1, 0.9, (32, 2)).astype(np.float32)
    wh = rng.uniform(0.01, 0.3, (32, 2)).astype(np.float32)
    for lb in (False, True):
        out["correct_boxes/letterbox%d" % int(lb)] = yolo_correct_boxes(
            xy.copy(), wh.copy(), [640, 640], np.array([540, 1024]), lb)
    out["correct_boxes/xy"], out["correct_boxes/wh"] = xy, wh

    attention_cases(block)
    att = {k: out.pop(k) for k in list(out) if k.startswith("block/att_") or k.startswith("attnet/")}
    np.savez_compressed(os.path.join(HERE, "attention_golden.npz"), **att)
    np.savez_compressed(os.path.join(HERE, "drone_golden.npz"), **out)
    with open(os.path.join(HERE, "shapes.json"), "w") as f:
        json.dump(shapes_all, f, separators=(",", ":"))
    print("wrote", len(out), "arrays;", os.path.getsize(os.path.join(HERE, "drone_golden.npz")), "bytes")


if __name__ == "__main__":
    main()
