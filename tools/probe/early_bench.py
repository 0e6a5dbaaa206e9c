#!/usr/bin/env python3
"""micro-benchmarks of the bandwidth-bound early layers: fused Focus+stem vs focus_pack + conv; the 1x1 / stride-2 /
small-channel 3x3 layers at the benchmark size, every variant; GB/s of algorithmic traffic."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from glsdet_amd._lib import GlsdetError
from glsdet_amd.engine import Engine

eng = Engine("f16")


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


img = torch.randn(8, 3, 800, 1344, device="cuda")
w = torch.randn(32, 12, 3, 3) / 10
pk = eng.pack_conv([(w, torch.ones(32), torch.zeros(32))], 16)
out = eng.tensor(8, 400, 672, 32)
packed = eng.tensor(8, 400, 672, 16)
t_f = timeit(lambda: eng.focus_conv(img, pk, "silu", out=out))
t_p = timeit(lambda: eng.focus_pack(img, out=packed))
t_c = timeit(lambda: eng.conv(packed, pk, 1, 1, "silu", out=out))
mb = (img.numel() * 4 + 8 * 400 * 672 * 32 * 2) / 1e6
print("stem: fused %.1f us (%.0f GB/s) | focus_pack %.1f + conv %.1f us" % (t_f, mb / t_f * 1e3, t_p, t_c))

SHAPES = [(8, 200, 336, 64, 64, 1, 1), (8, 200, 336, 32, 32, 1, 1), (8, 100, 168, 128, 128, 1, 1), (8, 100, 168, 256, 128, 1, 1),
          (8, 50, 84, 256, 256, 1, 1), (8, 400, 672, 32, 64, 3, 2), (8, 200, 336, 64, 128, 3, 2), (8, 200, 336, 32, 32, 3, 1)]
HINTS = {"auto": 0, "ws1x1": 3, "ring64k64": 10, "ring128k64": 11, "g128x128": (128 << 16) | 128, "g64x128": (64 << 16) | 128,
         "g64x64": (64 << 16) | 64, "g64x64k64": (64 << 16) | 64 | 0x8000, "g64x128k64": (64 << 16) | 128 | 0x8000, "g32x128": (32 << 16) | 128}
for (n, H, W, cin, cout, k, s) in SHAPES:
    x = eng.tensor(n, H, W, cin)
    x.buf.view(torch.float16).normal_()
    wt = torch.randn(cout, cin, k, k) / (cin * k * k) ** 0.5
    pkc = eng.pack_conv([(wt, torch.ones(cout), torch.zeros(cout))], cin)
    Ho, Wo = (H + s - 1) // s, (W + s - 1) // s
    mb = (n * H * W * cin + n * Ho * Wo * cout) * 2 / 1e6
    line = "%dx%d s%d %4d->%4d @%dx%d (%.0f MB): " % (k, k, s, cin, cout, H, W, mb)
    for name, h in HINTS.items():
        try:
            o = eng.conv(x, pkc, s, (k - 1) // 2, "silu", tile_hint=h)
        except GlsdetError:
            continue
        us = timeit(lambda: eng.conv(x, pkc, s, (k - 1) // 2, "silu", out=o, tile_hint=h))
        line += "%s %.1f (%.0f) " % (name, us, mb / us * 1e3)
    print(line, flush=True)
