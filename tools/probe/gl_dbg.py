import sys; sys.path.insert(0, '/root/repo')
import torch, numpy as np
from oracle import glsdet_oracle as O, mpdet_oracle as M
from tests.helpers import calibrated_resdet_sd
from tests.test_hip_ops import _to_view
from glsdet_amd.engine import Engine
from glsdet_amd.resdet import ResDetBuilder
x = O.synth_input((1, 3, 128, 160), 7)
sd = calibrated_resdet_sd("mpdet", 1, x, gl_fusion=True)
stages = M.resnet(sd, "backbone", x)
for mode in ("f32", "f16"):
    eng = Engine(mode)
    b = ResDetBuilder(eng, sd)
    for i in (1, 2, 3):
        f = stages[i]
        r = (lambda t: t.half().float()) if mode == "f16" else (lambda t: t)
        p = "neck.gl_fusion.%d" % i
        want = r(f) + O.patch_conv_nonlocal_new(sd, p, r(f))
        for assoc in ("re", "dir"):
            out = b.gl_fusion(p, _to_view(eng, f), assoc)
            torch.cuda.synchronize()
            got = out.to_nchw().cpu()
            print(mode, "level", i, tuple(f.shape), assoc, "max|x| %.1f max|want| %.1f err %.3e nan %s inf %s" % (
                float(f.abs().max()), float(want.abs().max()), float((got - want).abs().max()), bool(torch.isnan(got).any()), bool(torch.isinf(got).any())))
