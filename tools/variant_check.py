#!/usr/bin/env python3
"""Every conv of the benchmark detectors on every kernel variant that accepts it, at the benchmark's own size
and on the benchmark's own data, LAYER BY LAYER: while the detector is emitted eagerly, each conv is first run on
the variant under test into a scratch tensor (same operands) and then by the default selection; the two results
are compared element by element (Engine.shadow).  Finds a variant that is wrong on a shape / view the unit tests
do not contain (round 1: two drift-dependent LDS races of the hand-synchronised halo kernels).

Per-layer bars.  exact-f32 mode: the variants compute the same fp32 sums in another order -- 5e-5 x max|out|.
f16 mode (the benchmarked instantiations, incl. the LDS-DMA ring kernels): the same fp32 sums rounded once to
fp16, so two correct variants differ by at most one fp16 ulp where the rounding flips -- 1.2e-3 x max|out|
(2^-10 = 9.8e-4).  A race that corrupts a tile is orders of magnitude above either.

Why not compare logits in f16: the random-weight benchmark net at 800x1344 amplifies ONE flipped fp16 rounding
to several percent of max|logit| (r02: every generic tile vs the default selection 4.2e-2, both correct -- in
exact-f32 mode the same pairs agree to 1.6e-4).  `--whole` still prints that table.

usage: variant_check.py [f32] [f16] [workload ...] [--whole]
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import argparse
import torch
import bench

HINTS = {"generic": 1, "halo": 2, "halo_wp": 4, "halo_co64": 5, "halo_ring64": 8, "halo_ring128": 9,
         "halo_ring64k64": 10, "halo_ring128k64": 11, "halo_ring8": 12, "halo_ring8k64": 13, "ws1x1": 3, "g128x128": (128 << 16) | 128, "g64x128": (64 << 16) | 128,
         "g64x64": (64 << 16) | 64, "g32x128": (32 << 16) | 128, "g64x64k64": (64 << 16) | 64 | 0x8000,
         "g64x128k64": (64 << 16) | 128 | 0x8000, "g128x128k64": (128 << 16) | 128 | 0x8000, "g128x256": (128 << 16),
         "g128x256k64": (128 << 16) | 0x8000,
         "ring64_10x12": 0x108, "ring128_10x12": 0x109, "ring64k64_10x12": 0x10a, "ring128k64_10x12": 0x10b, "ring8k64_10x24": 0x10d,
         "ring64_6x21": 0x208, "ring128_6x21": 0x209, "ring64k64_6x21": 0x20a, "ring128k64_6x21": 0x20b, "ring8k64_6x42": 0x20d,
         "gemm_p64x64": 16, "gemm_p128x128k64": 20, "gemm_pw64x128": 22, "gemm_pw64x64": 24, "gemm_pw32x128": 25, "gemm1_64x64": 29}
MULTI = {"mring64_10x12": 0x108, "mring64k64_10x12": 0x10a, "mring64_6x21": 0x208, "mring64k64_6x21": 0x20a, "mring128k64_6x21": 0x20b,
         "m128x128": (128 << 16) | 128, "m64x128": (64 << 16) | 128, "m64x64": (64 << 16) | 64,
         "m64x64k64": (64 << 16) | 64 | 0x8000, "m64x128k64": (64 << 16) | 128 | 0x8000, "m128x128k64": (128 << 16) | 128 | 0x8000}
dev = "cuda:0"
BAR = {"f32": 5e-5, "f16": 1.2e-3}
args = argparse.Namespace(dtype="f16", conf=0.25, candidates=2000)


def _inputs(workload):
    kind, tag, H, W, bs = bench.WORKLOADS[workload]
    img = torch.randn(bs, 3, H, W, generator=torch.Generator(device=dev).manual_seed(0), device=dev)
    if kind in bench.RESDET:
        from glsdet_amd.synth import synth_input, synth_resdet_state_dict
        extra = dict(gl_fusion=True) if kind == "mpdet_gl" else {}
        return kind, synth_resdet_state_dict("mpdet" if kind == "mpdet_gl" else kind, 0, synth_input((1, 3, 128, 160), 100), **extra), img
    return kind, bench.synthetic_state_dict(tag), img


def emit(workload, dtype, shadow=None):
    """Emit the detector eagerly on a fresh engine (optionally in shadow mode) -> (raw head outputs, shadow log)."""
    from glsdet_amd.engine import Engine
    kind, sd, img = _inputs(workload)
    eng = Engine(dtype, dev)
    eng.shadow = shadow
    if kind in bench.RESDET:
        from glsdet_amd.resdet import HipGflDetector
        det = HipGflDetector("mpdet" if kind == "mpdet_gl" else kind, sd, dtype=dtype, device=dev)
        c, r = det._emit(eng, img)
        outs = [v.to_nchw() for v in list(c) + list(r)]
    else:
        from glsdet_amd.nets import build_forward
        lv, nc, _ = build_forward(kind, eng, sd, img)
        outs = [v.to_nchw(5 + nc) for v in lv]
    torch.cuda.synchronize()
    return outs, (shadow or {}).get("log", [])


def check(workloads, out=print, dtype="f16", whole=False):
    """-> number of (workload, variant) pairs with a layer beyond BAR[dtype] x max|out| of the default selection."""
    bad = 0
    for wl in workloads:
        ref = emit(wl, dtype)[0] if whole else None
        for key, table in (("hint", HINTS), ("multi", MULTI)):
            for name, h in table.items():
                outs, log = emit(wl, dtype, {"hint": h if key == "hint" else None, "multi": h if key == "multi" else None, "log": []})
                if not log:
                    out("%-30s %s %-16s accepts no conv of this detector" % (wl, dtype, name))
                    continue
                worst = max(log, key=lambda e: (e["nan"], e["err"] / max(e["scale"], 1e-30)))
                rel = worst["err"] / max(worst["scale"], 1e-30)
                differ = sum(e["differ"] for e in log) / len(log)
                flag = "  <-- SUSPECT" if (worst["nan"] or rel > BAR[dtype]) else ""
                bad += bool(flag)
                line = "%-30s %s %-16s %3d convs, worst layer %.1e x max|out| (n,h,w,cin,cout,k,s = %s), %.2f %% of the elements differ%s" % (
                    wl, dtype, name, len(log), rel, worst["shape"], 100 * differ, flag)
                if whole:
                    scale = max(float(r.abs().max()) for r in ref)
                    line += "  | logits of the default selection reproduced to %.1e" % (
                        max(float((g - r).abs().max()) for g, r in zip(outs, ref)) / scale)
                out(line)
    return bad


if __name__ == "__main__":
    wls = [a for a in sys.argv[1:] if a in bench.WORKLOADS] or ["yolox_s_glfusion_1344x800_bs8", "mp_det_res50_1344x800_bs8"]
    n = sum(check(wls, lambda s: print(s, flush=True), dt, "--whole" in sys.argv)
            for dt in ([a for a in sys.argv[1:] if a in BAR] or ["f32", "f16"]))
    print("suspect variants:", n)
    sys.exit(1 if n else 0)
