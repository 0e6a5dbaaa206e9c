#!/usr/bin/env python3
"""diagnostic: the persistent 1x1 kernel (tile_hint 16..26) on a few hand-picked problems, error per variant"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch, torch.nn.functional as F
from glsdet_amd._lib import GlsdetError
from glsdet_amd.engine import Engine
from tests.test_hip_ops import _to_view

CASES = [  # n, cin, cout, h, w, res
    (1, 64, 64, 16, 16, 0), (1, 64, 64, 25, 38, 0), (2, 128, 128, 30, 40, 0), (1, 8, 136, 25, 38, 0), (1, 8, 136, 25, 38, 1),
    (1, 64, 64, 25, 38, 1), (2, 256, 256, 20, 33, 0), (1, 32, 32, 40, 50, 0), (1, 640, 256, 20, 21, 0), (2, 72, 40, 9, 11, 2),
]
for mode in ("f16", "f32"):
    eng = Engine(mode)
    for (n, cin, cout, h, w, res) in CASES:
        g = torch.Generator().manual_seed(cin + cout)
        x = torch.randn(n, cin, h, w, generator=g)
        wt = torch.randn(cout, cin, 1, 1, generator=g) / np.sqrt(cin)
        sc, bi = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.3
        r = (lambda t: t.half().float()) if mode == "f16" else (lambda t: t)
        pre = F.conv2d(r(x), r(wt)) * sc[None, :, None, None] + bi[None, :, None, None]
        rt = torch.randn(pre.shape, generator=g) if res else None
        ref = F.silu(pre) if res == 0 else (F.silu(pre) + r(rt) if res == 1 else F.silu(pre + r(rt)))
        pk = eng.pack_conv([(wt, sc, bi)], cin)
        line = "%s n%d ci%d co%d %dx%d r%d:" % (mode, n, cin, cout, h, w, res)
        for hint in [1] + list(range(16, 32)):
            try:
                out = eng.conv(_to_view(eng, x), pk, 1, 0, "silu", res=_to_view(eng, rt) if res else None, tile_hint=hint, res_first=(res == 2))
            except GlsdetError:
                line += " %d:-" % hint
                continue
            torch.cuda.synchronize()
            got = out.to_nchw(cout).cpu()
            d = (got - ref).abs()
            bad = (d > 1e-2 * max(1.0, float(ref.abs().max())))
            line += " %d:%.1e" % (hint, float(d.max()))
            if bad.any() and hint == 16:
                idx = bad.nonzero()
                line += "[bad %d of %d; first %s; co range %d-%d px rows %d-%d]" % (int(bad.sum()), bad.numel(), idx[0].tolist(), int(idx[:, 1].min()), int(idx[:, 1].max()), int(idx[:, 2].min()), int(idx[:, 2].max()))
        print(line, flush=True)
