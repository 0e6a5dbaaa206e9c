#!/usr/bin/env python3
"""Replay the benchmark plan many times (two instances in flight) and verify that every replay
reproduces the first one bit for bit and never raises the NMS overflow flag."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import argparse

import torch

import bench
from glsdet_amd.detector import HipDetector

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--no-autotune", action="store_true")
ap.add_argument("--conf", type=float, default=0.25)
ap.add_argument("--candidates", type=int, default=2000)
ap.add_argument("--dtype", default="f16")
args = ap.parse_args()
dev = "cuda:0"
kind, tag, H, W, bs = bench.WORKLOADS["yolox_s_glfusion_1344x800_bs8"]
img = torch.randn(bs, 3, H, W, generator=torch.Generator(device=dev).manual_seed(0), device=dev)
sd = bench.calibrate_objectness(bench.synthetic_state_dict(tag), kind, img, args, dev)
det = HipDetector(kind, sd, dtype="f16", device=dev, autotune=not args.no_autotune)
post = dict(conf_thres=0.25, nms_thres=0.65, max_det=3000)
cs = [det.compile(bs, H, W, post, use_graph=True, instance=i) for i in range(2)]
for c in cs:
    c.img.copy_(img)
torch.cuda.synchronize()
ref = {}
bad = 0
for step in range(args.steps):
    for c in cs:
        HipDetector.run_async(c)
    torch.cuda.synchronize()
    for i, c in enumerate(cs):
        st = int(c.nmsb["status"].item())
        cur = (c.nmsb["count"].clone(), c.nmsb["dets"].clone(), [l.to_nchw().clone() for l in c.levels])
        r = ref.setdefault(i, cur)        # instances tune separately: compare each with its own first replay
        same_levels = [bool(torch.equal(a, b)) for a, b in zip(cur[2], r[2])]
        ok = st == 0 and torch.equal(cur[0], r[0]) and torch.equal(cur[1], r[1]) and all(same_levels)
        if not ok:
            bad += 1
            print("step %d instance %d: status %d count_equal %s dets_equal %s levels_equal %s cnt %s" % (
                step, i, st, bool(torch.equal(cur[0], r[0])), bool(torch.equal(cur[1], r[1])), same_levels,
                c.nmsb["ws"][: 4 * bs].view(torch.int32).tolist()), flush=True)
            if bad > 10:
                sys.exit(1)
print("done: %d steps, %d bad replays" % (args.steps, bad))
sys.exit(1 if bad else 0)
