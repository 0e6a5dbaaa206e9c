#!/usr/bin/env python3
"""Micro-benchmark: n identical-shape convs launched one by one (each at its tuned best) vs one
glsdet_conv2d_multi launch."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from glsdet_amd.engine import Engine

CASES = [  # n_problems, (n, H, W, cin, cout, k, stride)
    (4, (8, 13, 21, 256, 128, 3, 1)), (4, (8, 25, 42, 256, 128, 3, 1)), (4, (8, 50, 84, 128, 64, 3, 2)),
    (4, (8, 25, 42, 64, 192, 1, 1)), (2, (8, 50, 42, 128, 128, 3, 1)), (2, (8, 50, 84, 128, 128, 3, 1)),
    (2, (8, 25, 42, 128, 128, 3, 1)), (2, (8, 100, 168, 128, 128, 3, 1)), (4, (8, 25, 42, 64, 64, 3, 1)),
    (2, (8, 13, 21, 256, 256, 3, 1)),
]
HINTS = {"auto": 0, "64x64": (64 << 16) | 64, "64x64k64": (64 << 16) | 64 | 0x8000, "64x128": (64 << 16) | 128,
         "128x128": (128 << 16) | 128, "128x128k64": (128 << 16) | 128 | 0x8000}


def timeit(fn, reps=30):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    tuned = Engine("f16", autotune=True)
    eng = Engine("f16")
    for np_, (n, H, W, cin, cout, k, s) in CASES:
        xs, packs = [], []
        for i in range(np_):
            x = eng.tensor(n, H, W, cin)
            x.buf.view(torch.float16).normal_()
            w = torch.randn(cout, cin, k, k) / (cin * k * k) ** 0.5
            xs.append(x)
            packs.append(eng.pack_conv([(w, torch.ones(cout), torch.zeros(cout))], cin))
        outs = [tuned.conv(x, pk, s, k // 2, "silu") for x, pk in zip(xs, packs)]
        sep = timeit(lambda: [tuned.conv(x, pk, s, k // 2, "silu", out=o) for x, pk, o in zip(xs, packs, outs)])
        line = "%d x [%dx%d s%d %d->%d @%dx%d]: separate(tuned) %.1fus | multi: " % (np_, k, k, s, cin, cout, H, W, sep)
        for name, h in HINTS.items():
            try:
                eng.conv_multi(xs, packs, s, k // 2, "silu", outs=outs, tile_hint=h)
            except Exception:
                continue
            line += "%s %.1f | " % (name, timeit(lambda: eng.conv_multi(xs, packs, s, k // 2, "silu", outs=outs, tile_hint=h)))
        print(line, flush=True)


if __name__ == "__main__":
    main()
