#!/usr/bin/env python3
"""GPU micro-benchmark: every conv kernel variant (tile_hint) on the layer shapes that
dominate the bench workload.  Prints microseconds and algorithmic TFLOP/s per variant."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from glsdet_amd._lib import GlsdetError
from glsdet_amd.engine import Engine

SHAPES = [  # n, H, W, cin, cout, k, stride
    (8, 100, 168, 128, 128, 7, 1),
    (8, 50, 84, 256, 256, 5, 1),
    (8, 100, 168, 128, 128, 3, 1),
    (8, 100, 168, 128, 256, 3, 1),
    (8, 50, 84, 128, 128, 3, 1),
    (8, 25, 42, 512, 512, 3, 1),
    (8, 100, 168, 64, 64, 3, 1),
    (8, 100, 168, 128, 128, 1, 1),
    (8, 200, 336, 64, 64, 1, 1),
    (8, 50, 84, 256, 256, 1, 1),
    (8, 50, 84, 768, 256, 1, 1),
    (8, 200, 336, 64, 128, 3, 2),
    (8, 400, 672, 32, 64, 3, 2),
    (8, 400, 672, 16, 32, 3, 1),
    (8, 200, 336, 32, 32, 3, 1),
]
GLS_1X1 = [  # the 1x1 layers of YOLOX-s + GL neck at 8 x 800 x 1344 (profiles/r02_l/ops.tsv)
    (8, 200, 336, 64, 64, 1, 1), (8, 200, 336, 32, 32, 1, 1), (8, 100, 168, 128, 128, 1, 1), (8, 100, 168, 256, 128, 1, 1),
    (8, 50, 84, 256, 256, 1, 1), (8, 50, 84, 128, 128, 1, 1), (8, 50, 84, 256, 128, 1, 1), (8, 50, 84, 512, 256, 1, 1),
    (8, 50, 84, 640, 256, 1, 1), (8, 50, 84, 384, 256, 1, 1), (8, 25, 42, 512, 256, 1, 1), (8, 25, 42, 1024, 512, 1, 1),
    (8, 25, 42, 512, 512, 1, 1), (8, 25, 42, 256, 256, 1, 1),
]
QUANT = [  # tile-count quantisation probe: the same layer on maps that need 1.2 / 0.94 / 0.75 rounds of workgroups
    (8, 100, 168, 128, 128, 7, 1), (8, 80, 168, 128, 128, 7, 1), (8, 64, 168, 128, 128, 7, 1), (8, 104, 192, 128, 128, 7, 1),
    (8, 50, 84, 256, 256, 5, 1), (8, 48, 80, 256, 256, 5, 1), (8, 40, 80, 256, 256, 5, 1),
    (8, 100, 168, 128, 256, 3, 1), (8, 80, 160, 128, 256, 3, 1),
]
RESNET = [  # ResNet-50 / FPN / head layers at 8 x 800 x 1344
    (8, 50, 84, 1024, 256, 1, 1),
    (8, 50, 84, 256, 1024, 1, 1),
    (8, 100, 168, 512, 128, 1, 1),
    (8, 100, 168, 128, 512, 1, 1),
    (8, 200, 336, 64, 256, 1, 1),
    (8, 25, 42, 2048, 512, 1, 1),
    (8, 25, 42, 512, 2048, 1, 1),
    (8, 50, 84, 256, 256, 3, 1),
    (8, 25, 42, 512, 512, 3, 1),
    (8, 100, 168, 256, 256, 3, 1),
    (8, 100, 168, 128, 128, 3, 1),
]
HINTS = {"auto": 0, "halo": 2, "halowp": 4, "halo64": 5, "dma64": 6, "dma128": 7, "ring64": 8, "ring128": 9, "ring64k64": 10, "ring128k64": 11, "ring8": 12, "ring8k64": 13, "ws1x1": 3, "g128x128": (128 << 16) | 128, "g64x128": (64 << 16) | 128,
         "g64x64": (64 << 16) | 64, "g64x64k64": (64 << 16) | 64 | 0x8000, "g64x128k64": (64 << 16) | 128 | 0x8000, "g32x128": (32 << 16) | 128,
         "g128x128k64": (128 << 16) | 128 | 0x8000, "g128x256": (128 << 16), "g128x256k64": (128 << 16) | 0x8000,
         "p64x64": 16, "p128x64": 17, "p64x128": 18, "p128x128": 19, "p128x128k64": 20, "p32x128": 21, "pw64x128": 22, "pw128x64": 23,
         "pw64x64": 24, "pw32x128": 25, "pw128x128": 26, "p64x64k64": 27, "p128x64k64": 28, "s64x64": 29, "s64x128": 30, "s32x128": 31,
         "ring64g1": 0x108, "ring128g1": 0x109, "ring64k64g1": 0x10a, "ring128k64g1": 0x10b, "ring8k64g1": 0x10d,
         "ring64g2": 0x208, "ring128g2": 0x209, "ring64k64g2": 0x20a, "ring128k64g2": 0x20b, "ring8k64g2": 0x20d}


def main():
    args = sys.argv[1:]
    use_res = "res" in args
    args = [a for a in args if a != "res"]
    shapes = SHAPES
    if args and args[0] == "resnet":
        shapes, args = RESNET, args[1:]
    elif args and args[0] == "quant":
        shapes, args = QUANT, args[1:]
    elif args and args[0] == "gls1x1":
        shapes, args = GLS_1X1, args[1:]
    only = args or None
    eng = Engine("f16")
    for (n, H, W, cin, cout, k, s) in shapes:
        x = eng.tensor(n, H, W, cin)
        x.buf.view(torch.float16).normal_()
        w = torch.randn(cout, cin, k, k) / (cin * k * k) ** 0.5
        pk = eng.pack_conv([(w, torch.ones(cout), torch.zeros(cout))], cin)
        flops = 2.0 * n * ((H + s - 1) // s) * ((W + s - 1) // s) * cout * cin * k * k
        line = "%dx%d s%d %4d->%4d @%dx%d: " % (k, k, s, cin, cout, H, W)
        for name, h in HINTS.items():
            if only and name not in only:
                continue
            try:
                out = eng.conv(x, pk, s, (k - 1) // 2, "silu", tile_hint=h)
            except GlsdetError:
                continue
            res = None
            if use_res:                     # residual layer (ResNet conv3: add, then ReLU)
                res = eng.tensor(out.n, out.h, out.w, out.c)
                res.buf.view(torch.float16).normal_()
            conv = lambda: eng.conv(x, pk, s, (k - 1) // 2, "relu" if use_res else "silu", out=out, tile_hint=h, res=res,
                                    res_first=use_res)
            # 20 launches recorded into a plan and replayed as ONE hipGraph: an eager python loop is host bound below
            # ~10 us per launch; the figure includes the ~1.5 us boundary between two dependent kernels
            reps = 20
            plan = eng.new_plan()
            with plan:
                for _ in range(reps):
                    conv()
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                plan.run(st)
                st.synchronize()
                plan.capture(st)
                plan.launch(st)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(st)
                for _ in range(5):
                    plan.launch(st)
                e1.record(st)
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / (5 * reps)
            line += "%s %.1fus %.0fTF | " % (name, us, flops / us / 1e6)
        print(line, flush=True)


if __name__ == "__main__":
    main()
