#!/usr/bin/env python3
"""Where does the f16 mode's end-to-end logit error come from?  CPU only (oracle emulation).

For a whole-model golden case: (1) per-tensor trace -- relative rms error of every stored tensor of the
fp16-storage emulation against the fp32 oracle, in execution order; (2) attribution -- the logit error
when ONLY one group of tensors is rounded to fp16 (everything else fp32), and when everything BUT that
group is rounded.  Prints the table DESIGN.md section 4c quotes.

    python tools/f16_attribution.py [gl_s_seed0 base_s_seed0 ...]
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import glsdet_oracle as O          # noqa: E402  (a tool of the test infrastructure)
from tests.helpers import model_case           # noqa: E402


def group_of(name: str) -> str:
    if name == "input":
        return "input"
    if name.endswith(".weight"):
        return "weights"
    for g in ("backbone.backbone.stem", "backbone.backbone.dark2", "backbone.backbone.dark3", "backbone.backbone.dark4",
              "backbone.backbone.dark5", "backbone.Patch_conv_feat1", "backbone.Patch_conv_feat2", "backbone.P3_Identity",
              "backbone.P4_Identity", "backbone.P5_Identity", "head.stems", "head.cls_convs", "head.reg_convs"):
        if name.startswith(g):
            return g
    if name.startswith("backbone."):
        return "backbone.pafpn"
    return "other"


def run(tag):
    golden = np.load(os.path.join(ROOT, "tests", "golden", "drone_golden.npz"))
    with open(os.path.join(ROOT, "tests", "golden", "shapes.json")) as f:
        shapes = json.load(f)
    meta, sd, x, outs, _ = model_case(golden, shapes, tag)
    fwd = O.FORWARDS[meta["model"]]
    scale = max(float(o.abs().max()) for o in outs)

    def err(o):
        d = torch.cat([(a - b).flatten() for a, b in zip(o, outs)])
        return float(d.abs().max()) / scale, float(d.pow(2).mean().sqrt()) / scale

    with torch.no_grad():
        O.TRACE = ref = {}
        fwd(sd, x)
        O.TRACE = emu = {}
        with O.fp16_storage():
            full = fwd(sd, x)
        O.TRACE = None
        print("== %s: max|logit| %.3f; fp16-storage emulation vs fp32 reference: max %.4f rms %.5f (x max|logit|)"
              % (tag, scale, *err(full)))
        print("-- per-tensor trace (execution order): rel rms err, max|value|")
        for name in emu:
            a, b = emu[name], ref[name]
            rel = float((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt().clamp(min=1e-12))
            print("  %-58s %.2e  %9.2f" % (name, rel, float(b.abs().max())))
        groups = sorted({group_of(n) for n in emu} | {"input", "weights"})
        print("-- attribution: logit error (max, rms; x max|logit|) rounding ONLY the group / everything BUT the group")
        for g in groups:
            with O.fp16_storage(lambda n, g=g: group_of(n) == g):
                only = err(fwd(sd, x))
            with O.fp16_storage(lambda n, g=g: group_of(n) != g):
                but = err(fwd(sd, x))
            print("  %-28s only: %.4f %.5f   all-but: %.4f %.5f" % (g, *only, *but))


if __name__ == "__main__":
    torch.set_num_threads(8)
    for t in (sys.argv[1:] or ["gl_s_seed0", "base_s_seed0"]):
        run(t)
