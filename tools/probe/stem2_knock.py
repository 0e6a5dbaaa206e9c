#!/usr/bin/env python3
"""Knock-out timing of focus_stem_down (GLSDET_STEM2_DBG bits: 1 no image loads, 2 no patch scatter, 4 no phase-A MFMAs,
8 no P1 epilogue, 16 no phase-B MFMAs, 32 no store).  Results of dbg != 0 runs are garbage by design."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from glsdet_amd.engine import Engine
from glsdet_amd.arch import _Table
from glsdet_amd.nets import NetBuilder
from glsdet_amd.synth import synth_state_dict

eng = Engine("f16")
t = _Table(); t.conv_bn("m.conv", 12, 32, 3); t.conv_bn("d", 32, 64, 3)
sd = synth_state_dict(t, 6)
b = NetBuilder(eng, sd)
p1, p2 = b._pack("m.conv", [b._bn_part("m.conv")], 16), b._pack("d", [b._bn_part("d")], 32)
x = torch.randn(8, 3, 800, 1344, device="cuda")
out = eng.tensor(8, 200, 336, 64)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(torch.cuda.current_stream())
    for _ in range(n): fn()
    e1.record(torch.cuda.current_stream()); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1000
eng.stream = torch.cuda.current_stream()
for dbg in [0, 1, 2, 3, 4, 8, 12, 16, 32, 48, 60, 63]:
    os.environ["GLSDET_STEM2_DBG"] = str(dbg)
    print("dbg %2d: %.1f us" % (dbg, timeit(lambda: eng.focus_conv_down(x, p1, "silu", p2, "silu", out=out))))
os.environ["GLSDET_STEM2_DBG"] = "0"
mid = eng.tensor(8, 400, 672, 32)
print("focus_conv: %.1f us" % timeit(lambda: eng.focus_conv(x, p1, "silu", out=mid)))
print("conv s2   : %.1f us" % timeit(lambda: eng.conv(mid, p2, 2, 1, "silu", out=out)))
