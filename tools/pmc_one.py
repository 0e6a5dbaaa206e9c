#!/usr/bin/env python3
"""Run ONE conv shape/variant a few times (for rocprofv3 --pmc runs).
usage: pmc_one.py n H W cin cout k stride hint_name"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from glsdet_amd.engine import Engine
from tools.conv_variants import HINTS
n, H, W, cin, cout, k, s = map(int, sys.argv[1:8])
h = HINTS[sys.argv[8]]
eng = Engine("f16")
x = eng.tensor(n, H, W, cin)
x.buf.view(torch.float16).normal_()
w = torch.randn(cout, cin, k, k) / (cin * k * k) ** 0.5
pk = eng.pack_conv([(w, torch.ones(cout), torch.zeros(cout))], cin)
out = eng.conv(x, pk, s, (k - 1) // 2, "silu", tile_hint=h)
for _ in range(5):
    eng.conv(x, pk, s, (k - 1) // 2, "silu", out=out, tile_hint=h)
torch.cuda.synchronize()
