#!/usr/bin/env python3
"""Throughput of the device preprocessing (8 UAVDT-sized frames -> 8x3x640x640 and 8x3x800x1344) next to
the reference's CPU path (PIL + numpy, one thread, as drone/yolo.py runs it)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from glsdet_amd.preprocess import DronePreprocessor
from oracle import preprocess_oracle as P
from tests.test_preprocess import synth_image

imgs = [synth_image((540, 1024), i) for i in range(8)]
dev_imgs = [torch.from_numpy(i).cuda() for i in imgs]
p = DronePreprocessor()
for shape in ((640, 640), (800, 1344)):
    out = p(dev_imgs, shape, True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        p(dev_imgs, shape, True, out=out)
    torch.cuda.synchronize()
    gpu = (time.perf_counter() - t0) / 20
    t0 = time.perf_counter()
    for im in imgs:
        P.drone_preprocess(im, shape, True)
    cpu = time.perf_counter() - t0
    print("8 x 540x1024 -> %dx%d letterbox: device %.3f ms/batch (%.0f img/s, frames already in HBM), "
          "PIL+numpy on the host %.1f ms/batch (%.0f img/s)" % (shape[0], shape[1], gpu * 1e3, 8 / gpu, cpu * 1e3, 8 / cpu))
