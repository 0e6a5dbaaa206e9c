#!/usr/bin/env python3
"""One conv layer at benchmark size in exact-f32 mode: the outputs of several kernel variants against the generic
kernel (differences beyond fp32 summation-order noise mean a wrong variant).  usage: hint_diff.py [hint ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from glsdet_amd.engine import Engine

hints = [int(h) for h in sys.argv[1:]] or [2, 5, 8, 9, 10]
eng = Engine(os.environ.get("HINT_DIFF_DTYPE", "f32"))
for (n, H, W, cin, cout, k) in [(8, 100, 168, 128, 256, 3), (8, 50, 84, 256, 256, 5), (8, 100, 168, 128, 128, 3), (4, 37, 53, 320, 136, 3)]:
    x = eng.tensor(n, H, W, cin)
    (x.buf.view(torch.float32)[: n * H * W * cin] if eng.dt else x.buf.view(torch.float16)[: n * H * W * cin]).normal_()
    w = torch.randn(cout, cin, k, k) / (cin * k * k) ** 0.5
    pk = eng.pack_conv([(w, torch.rand(cout) + 0.5, torch.randn(cout) * 0.3)], cin)
    ref = eng.conv(x, pk, 1, k // 2, "silu", tile_hint=1).to_nchw(cout).clone()
    line = []
    for h in hints:
        try:
            out = eng.conv(x, pk, 1, k // 2, "silu", tile_hint=h).to_nchw(cout)
            line.append("hint %d: %.2e" % (h, float((out - ref).abs().max())))
        except Exception as e:          # noqa: BLE001
            line.append("hint %d: n/a" % h)
    print("%dx%d %d->%d @%dx%d  max|ref| %.2f  " % (k, k, cin, cout, H, W, float(ref.abs().max())) + "  ".join(line), flush=True)
