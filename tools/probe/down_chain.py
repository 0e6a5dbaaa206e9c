#!/usr/bin/env python3
"""stride-2 3x3 conv + the CSPLayer's conv1|conv2: separate launches vs the chained launch (GLSDET_CHAIN_SKIP_Y), us per pair"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from glsdet_amd.engine import Engine

def timed(eng, fn, reps=20):
    plan = eng.new_plan()
    with plan:
        for _ in range(reps):
            fn()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        plan.run(st); st.synchronize(); plan.capture(st); plan.launch(st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(5):
            plan.launch(st)
        e1.record(st)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (5 * reps)

eng = Engine("f16")
for (n, H, W, cin, cout) in [(8, 400, 672, 32, 64), (8, 200, 336, 64, 128)]:
    x = eng.tensor(n, H, W, cin); x.buf.view(torch.float16).normal_()
    w1 = torch.randn(cout, cin, 3, 3) / (cin * 9) ** 0.5
    w2 = torch.randn(cout, cout, 1, 1) / cout ** 0.5
    p1 = eng.pack_conv([(w1, torch.ones(cout), torch.zeros(cout))], cin)
    p2 = eng.pack_conv([(w2, torch.ones(cout), torch.zeros(cout))], cout)
    y = eng.tensor(n, H // 2, W // 2, cout); y2 = eng.tensor(n, H // 2, W // 2, cout)
    line = "3x3 s2 %d->%d @%dx%d + 1x1 %d->%d:" % (cin, cout, H, W, cout, cout)
    for h1 in (0, (64 << 16) | 64, (64 << 16) | 128, (128 << 16) | 128, (64 << 16) | 64 | 0x8000, 10, 11):
        try:
            t1 = timed(eng, lambda: eng.conv(x, p1, 2, 1, "silu", out=y, tile_hint=h1))
            line += " conv[%x] %.1f" % (h1, t1)
        except Exception:
            pass
    for h2 in (0, (64 << 16) | 64, (64 << 16) | 64 | 0x8000, 24, 3):
        try:
            t2 = timed(eng, lambda: eng.conv(y, p2, 1, 0, "silu", out=y2, tile_hint=h2))
            line += " 1x1[%x] %.1f" % (h2, t2)
        except Exception:
            pass
    for skip in (False, True):
        ok = [True]
        def f():
            ok[0] = eng.conv_chain(x, p1, 2, 1, "silu", y, None, p2, "silu", 0, cout, y2, skip_y=skip) and ok[0]
        t = timed(eng, f)
        line += " chain(skip_y=%s) %.1f%s" % (skip, t, "" if ok[0] else " (declined)")
    print(line, flush=True)
