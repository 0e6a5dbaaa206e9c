"""GPU diagnostic: logit statistics and candidate counts of the bench workload."""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from glsdet_amd.detector import HipDetector
kind, tag, H, W, bs = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "yolox_s_glfusion_1344x800_bs8"]
sd = bench.synthetic_state_dict(tag)
for dtype in ("f16", "f32"):
    det = HipDetector(kind, sd, dtype=dtype)
    x = torch.randn(bs, 3, H, W, generator=torch.Generator(device="cuda").manual_seed(0), device="cuda")
    outs = det.forward_raw(x)
    for o in outs:
        o = o.cpu()
        print(dtype, tuple(o.shape), "reg std %.2f obj mean %.2f std %.2f cls mean %.2f std %.2f max %.1f nan %d" % (
            float(o[:, :4].std()), float(o[:, 4].mean()), float(o[:, 4].std()), float(o[:, 5:].mean()), float(o[:, 5:].std()), float(o.abs().max()), int(torch.isnan(o).sum())))
    score = torch.cat([(torch.sigmoid(o[:, 4:5]) * torch.sigmoid(o[:, 5:]).max(1, keepdim=True)[0]).flatten(1) for o in outs], 1)
    for thr in (0.1, 0.25, 0.5, 0.7, 0.9, 0.99):
        print(dtype, "score>=%.2f: per-image candidates" % thr, (score >= thr).sum(1).tolist())
