import torch
def timeit(fn, reps=30):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for mb in (17, 34, 69, 103, 137, 275):
    n = mb * 1000 * 1000 // 2
    x = torch.randn(n, device="cuda", dtype=torch.float16); y = torch.empty_like(x)
    us = timeit(lambda: y.copy_(x))
    us2 = timeit(lambda: torch.mul(x, 2.0, out=y))
    print("copy %d MB in + %d MB out: copy_ %.1f us (%.0f GB/s)  mul %.1f us (%.0f GB/s)" % (mb, mb, us, 2 * mb / us * 1e3, us2, 2 * mb / us2 * 1e3))
