#!/usr/bin/env python3
"""How much of the step is decode + NMS?  Replays the default workload with and without the
post-processing ops (three plan instances in flight, as bench.py)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import argparse
import torch
import bench
from glsdet_amd.detector import HipDetector

args = argparse.Namespace(dtype="f16", conf=0.25, candidates=2000)
dev = "cuda:0"
kind, tag, H, W, bs = bench.WORKLOADS["yolox_s_glfusion_1344x800_bs8"]
img = torch.randn(bs, 3, H, W, generator=torch.Generator(device=dev).manual_seed(0), device=dev)
sd = bench.calibrate_objectness(bench.synthetic_state_dict(tag), kind, img, args, dev)
det = HipDetector(kind, sd, dtype="f16", device=dev, autotune=True)
for name, post in (("with decode+NMS", dict(conf_thres=0.25, nms_thres=0.65, max_det=3000)), ("forward only", None)):
    cs = [det.compile(bs, H, W, post, use_graph=True, instance=i) for i in range(3)]
    for c in cs:
        c.img.copy_(img)
    torch.cuda.synchronize()
    for r in range(2):
        for i in range(10):
            HipDetector.run_async(cs[i % 3])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(60):
            HipDetector.run_async(cs[i % 3])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 60
    print("%s: %.4f ms/step, %.0f img/s" % (name, dt * 1e3, bs / dt))
