#!/usr/bin/env python3
"""Every conv of the benchmark detectors forced onto one kernel variant at a time (wherever that variant
accepts the problem): the raw logits must agree with the default selection to fp16 rounding.  Finds a
variant that is wrong on a shape / view the unit tests do not contain."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import argparse
import torch
import bench

HINTS = {"halo": 2, "halo_wp": 4, "halo_co64": 5, "halo_dma64": 6, "halo_dma128": 7, "halo_ring64": 8, "halo_ring128": 9, "halo_ring64k64": 10, "halo_ring128k64": 11, "ws1x1": 3, "g128x128": (128 << 16) | 128, "g64x128": (64 << 16) | 128,
         "g64x64": (64 << 16) | 64, "g32x128": (32 << 16) | 128, "g64x64k64": (64 << 16) | 64 | 0x8000,
         "g64x128k64": (64 << 16) | 128 | 0x8000, "g128x128k64": (128 << 16) | 128 | 0x8000}
MULTI = {"m128x128": (128 << 16) | 128, "m64x128": (64 << 16) | 128, "m64x64": (64 << 16) | 64,
         "m64x64k64": (64 << 16) | 64 | 0x8000, "m64x128k64": (64 << 16) | 128 | 0x8000, "m128x128k64": (128 << 16) | 128 | 0x8000}
dev = "cuda:0"
args = argparse.Namespace(dtype="f16", conf=0.25, candidates=2000)


def logits(workload):
    kind, tag, H, W, bs = bench.WORKLOADS[workload]
    img = torch.randn(bs, 3, H, W, generator=torch.Generator(device=dev).manual_seed(0), device=dev)
    if kind in bench.RESDET:
        from glsdet_amd.resdet import HipGflDetector
        from glsdet_amd.synth import synth_input, synth_resdet_state_dict
        sd = synth_resdet_state_dict(kind, 0, synth_input((1, 3, 128, 160), 100))
        c, r = HipGflDetector(kind, sd, dtype="f16", device=dev).forward_raw(img)
        return c + r
    from glsdet_amd.detector import HipDetector
    return HipDetector(kind, bench.synthetic_state_dict(tag), dtype="f16", device=dev).forward_raw(img)


def check(workloads, out=print):
    """-> number of (workload, variant) pairs whose logits leave fp16-rounding distance of the default."""
    bad = 0
    for wl in workloads:
        for k in ("GLSDET_FORCE_HINT", "GLSDET_FORCE_MULTI_HINT"):
            os.environ.pop(k, None)
        ref = logits(wl)
        scale = max(float(r.abs().max()) for r in ref)
        for env, table in (("GLSDET_FORCE_HINT", HINTS), ("GLSDET_FORCE_MULTI_HINT", MULTI)):
            for name, h in table.items():
                os.environ[env] = str(h)
                try:
                    got = logits(wl)
                finally:
                    os.environ.pop(env)
                err = max(float((g - r).abs().max()) for g, r in zip(got, ref))
                nan = any(bool(torch.isnan(g).any()) for g in got)
                flag = "  <-- SUSPECT" if (nan or err > 0.1 * scale) else ""
                bad += bool(flag)
                out("%-30s %-12s max|dlogit| %.4f (max |logit| %.2f)%s" % (wl, name, err, scale, flag))
    return bad


if __name__ == "__main__":
    n = check(sys.argv[1:] or ["yolox_s_glfusion_1344x800_bs8", "mp_det_res50_1344x800_bs8"], lambda s: print(s, flush=True))
    print("suspect variants:", n)
    sys.exit(1 if n else 0)
