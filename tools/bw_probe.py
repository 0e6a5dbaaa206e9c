#!/usr/bin/env python3
"""GPU probe: streaming copy bandwidth of the resample_copy kernel vs torch copy, to calibrate
what a memory-bound layer can reach."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from glsdet_amd.engine import Engine
eng = Engine("f16")
def timeit(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
for (n, h, w, c) in [(8, 100, 168, 128), (8, 200, 336, 64), (8, 400, 672, 32), (8, 50, 84, 256)]:
    x = eng.tensor(n, h, w, c); y = eng.tensor(n, h, w, c)
    nbytes = n * h * w * c * 2
    t1 = timeit(lambda: eng.resample(x, 1, out=y))
    a = torch.empty(nbytes // 2, dtype=torch.float16, device="cuda"); b = torch.empty_like(a)
    t2 = timeit(lambda: b.copy_(a))
    big = torch.empty(512 * 1024 * 1024 // 2, dtype=torch.float16, device="cuda"); big2 = torch.empty_like(big)
    t3 = timeit(lambda: big2.copy_(big), reps=5)
    print("%s %.1f MB: resample_copy %.1f us = %.2f TB/s | torch copy %.1f us = %.2f TB/s | torch 512MB copy %.2f TB/s" % (
        (n, h, w, c), nbytes / 1e6, t1, 2 * nbytes / t1 / 1e6, t2, 2 * nbytes / t2 / 1e6, 2 * 512 * 1.048576 / t3))
    del big, big2
