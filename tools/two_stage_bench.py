#!/usr/bin/env python3
"""Latency of the two-stage UFPMP-Det path for ONE 540x1024 frame (synthetic weights): coarse GFL at
1333x800 -> host packing -> device mosaic -> fine MPDet -> device back-mapping + merge NMS."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from glsdet_amd.resdet import HipGflDetector
from glsdet_amd.synth import synth_input, synth_resdet_state_dict
from glsdet_amd.ufp import TwoStagePipeline, UfpSecondStage, two_stage_detect
from glsdet_amd.ufp.packing import unified_foreground_packing
from tests.test_preprocess import synth_image

img = synth_image((540, 1024), 2)[:, :, ::-1].copy()
calib = synth_input((1, 3, 128, 160), 100)
coarse = HipGflDetector("gfl", synth_resdet_state_dict("gfl", 0, calib), dtype="f16", autotune=True)
fine = HipGflDetector("mpdet", synth_resdet_state_dict("mpdet", 1, calib), dtype="f16", autotune=True)
stage = UfpSecondStage()


def thr_for(det, x, keep):
    cls, _ = det.forward_raw(x)
    p = torch.sigmoid(torch.cat([c.flatten() for c in cls]))
    return float(torch.topk(p, keep).values[-1])


x1, _ = stage.pipeline_input(torch.from_numpy(img).cuda().contiguous())
c1 = dict(score_thr=thr_for(coarse, x1, 80), iou_thr=0.6, nms_pre=1000, max_per_img=100)
_, mid = two_stage_detect(coarse, fine, img, stage, c1, dict(score_thr=0.9999, iou_thr=0.6))
x2, _ = stage.pipeline_input(mid["canvas"])
c2 = dict(score_thr=thr_for(fine, x2, 600), iou_thr=0.6, nms_pre=1000, max_per_img=500)
for _ in range(3):
    merged, mid = two_stage_detect(coarse, fine, img, stage, c1, c2)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 20
for _ in range(n):
    merged, mid = two_stage_detect(coarse, fine, img, stage, c1, c2)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
boxes = mid["first"][0][:, :4]
t0 = time.perf_counter()
for _ in range(20):
    unified_foreground_packing(boxes.copy(), 1.5, [1024, 540])
tp = (time.perf_counter() - t0) / 20
print("two-stage, one 540x1024 frame: %.2f ms end to end (%.1f frames/s); host packing of %d boxes %.2f ms; "
      "%d chips, mosaic %dx%d, fine input %s, %d merged detections"
      % (dt * 1e3, 1 / dt, len(boxes), tp * 1e3, len(mid["chips"]), mid["canvas"].shape[1], mid["canvas"].shape[0],
         tuple(x2.shape), sum(len(m) for m in merged)))

frames = [synth_image((540, 1024), 100 + i)[:, :, ::-1].copy() for i in range(8)] * 4
pipes = {g: TwoStagePipeline(coarse, fine, stage, c1, c2, use_graph=g) for g in (False, True)}
for g in (False, True):                     # warm: plans (and graphs) for the mosaic shapes of these frames
    for f in frames[:8]:
        two_stage_detect(coarse, fine, f, stage, c1, c2, use_graph=g)
    pipes[g].run(frames[:8])
lanes = {w: TwoStagePipeline(coarse, fine, stage, c1, c2, workers=w) for w in (2, 3, 4)}
for w, pl in lanes.items():
    pl.run(frames[:8] * w)                  # every lane sees every shape once
torch.cuda.synchronize()
ref = None
for rep in range(2):
    line = []
    for name, fn in (("sequential eager", lambda: [two_stage_detect(coarse, fine, f, stage, c1, c2)[0] for f in frames]),
                     ("sequential graph replay", lambda: [two_stage_detect(coarse, fine, f, stage, c1, c2, use_graph=True)[0] for f in frames]),
                     ("two-stream pipeline eager", lambda: pipes[False].run(frames)),
                     ("two-stream pipeline graph replay", lambda: pipes[True].run(frames)),
                     ("2 lanes", lambda: lanes[2].run(frames)), ("3 lanes", lambda: lanes[3].run(frames)),
                     ("4 lanes", lambda: lanes[4].run(frames))):
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ref = out if ref is None else ref
        same = all(all(np.array_equal(a, b) for a, b in zip(x, y)) for x, y in zip(ref, out))
        line.append("%s %.1f%s" % (name, len(frames) / dt, "" if same else " (RESULTS DIFFER)"))
    print("32 frames, frames/s: " + "; ".join(line))
