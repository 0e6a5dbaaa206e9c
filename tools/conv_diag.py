#!/usr/bin/env python3
"""Timing experiments on the generic conv kernel: tile_hint diagnostic bits switch parts of the
kernel off (1 = no global loads after step 0, 2 = no LDS-read/MFMA, 4 = no epilogue).
usage: conv_diag.py [resnet]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from glsdet_amd.engine import Engine
from tools.conv_variants import HINTS, RESNET, SHAPES

CASES = [((8, 50, 84, 1024, 256, 1, 1), "g128x128k64"), ((8, 50, 84, 1024, 256, 1, 1), "g64x64"),
         ((8, 50, 84, 1024, 256, 1, 1), "g128x128"),
         ((8, 100, 168, 128, 512, 1, 1), "g128x128k64"), ((8, 25, 42, 512, 512, 3, 1), "g128x128"),
         ((8, 200, 336, 64, 128, 3, 2), "g128x128k64")]
NAMES = {0: "full", 1: "noload", 2: "nomma", 3: "noload+nomma", 4: "noepi", 5: "noload+noepi", 6: "nomma+noepi",
         7: "barriers only"}


def main():
    eng = Engine("f16")
    for (n, H, W, cin, cout, k, s), hn in CASES:
        x = eng.tensor(n, H, W, cin)
        x.buf.view(torch.float16).normal_()
        w = torch.randn(cout, cin, k, k) / (cin * k * k) ** 0.5
        pk = eng.pack_conv([(w, torch.ones(cout), torch.zeros(cout))], cin)
        line = "%dx%d s%d %4d->%4d @%dx%d %s: " % (k, k, s, cin, cout, H, W, hn)
        for dbg in range(8):
            h = HINTS[hn] | (dbg << 8)
            out = eng.conv(x, pk, s, (k - 1) // 2, "silu", tile_hint=h)
            for _ in range(3):
                eng.conv(x, pk, s, (k - 1) // 2, "silu", out=out, tile_hint=h)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                eng.conv(x, pk, s, (k - 1) // 2, "silu", out=out, tile_hint=h)
            e1.record()
            torch.cuda.synchronize()
            line += "%s %.1f | " % (NAMES[dbg], e0.elapsed_time(e1) * 1e3 / 20)
        print(line, flush=True)


if __name__ == "__main__":
    main()
