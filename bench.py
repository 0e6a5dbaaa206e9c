#!/usr/bin/env python3
"""Headline benchmark: images/sec of the detection forward pass (backbone -> GL-fusion PAFPN
-> decoupled head -> decode -> batched NMS) on MI355X, synthetic input, random-init weights.

    python bench.py --gpus N --steps K --warmup W

N>1 without a launcher (no WORLD_SIZE in the environment): this process starts N fresh child
processes -- one rank per GPU, before it has made any GPU call itself -- and rank 0 prints the line.
Under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` it is one of the ranks.

Workload (BASELINE.json: metric quoted at 1333x800 bs=8): YOLOX-s + GL-fusion neck
(`models.block.non_local.yolo_patch_nonlocal_plus`, the only GL-fusion detector the
reference wires up), nc=10, 8 images of 800x1344 (1333x800 keep-ratio, padded to /32) per
GPU, fp16 storage / fp32 accumulate.  A step = one batch through the captured hipGraph
(input already resident in HBM; detection counts read back to the host for every batch) + the
all_gather of the exchange record when N>1.  Weak scaling: 8 images per GPU, image i of the
global batch on rank i % N.

Prints ONE JSON line (rank 0).  Extra objects: `roofline` for the dominant kernel family
(conv family, per-launch numbers from HIP events around every op of an eager replay on the
launch stream; `frac` = that single-stream figure, `frac_end_to_end` = the same FLOPs over the
wall clock of the timed loop with all batches in flight), `cpu_baseline` (the CPU oracle, bounded
sample, rank 0 at N=1 only) and `secondary` = BASELINE config 3 (`mp_det_res50`: ResNet-50 + GL-fusion
plug-in + FPN + MPHead at 8 x 800 x 1344) measured the same way in the same run.
"""
import argparse
import glob
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F16_TFLOPS = 2500.0      # MI355X dense fp16 MFMA (MI355X_MICROARCH.md, chip-level parameters)
PEAK_F32_TFLOPS = 157.3       # exact-f32 MFMA (same table)
PEAK_HBM_GBS = 8000.0

# algorithmic GMAC per image (forward hooks on the imported reference):
# (conv GMAC, matmul GMAC).  roofline.achieved is computed from THESE, not from the
# (slightly larger, channel-padded) work the kernels execute.
ALGORITHMIC_GMAC = {
    "yolox_s_glfusion_1344x800_bs8": (61.527, 0.565),
    "yolox_s_glfusion_1344x800_bs4": (61.527, 0.565),
    "yolox_s_glfusion_1344x800_bs16": (61.527, 0.565),
    "yolox_s_glfusion_640x640_bs8": (23.439, 0.082),
    "yolox_s_base_640x640_bs8": (13.268, 0.0),
}

WORKLOADS = {
    # name: (detector kind, golden tag carrying the calibrated BN stats, H, W, per-GPU batch)
    "yolox_s_glfusion_1344x800_bs8": ("gl", "gl_s_seed0", 800, 1344, 8),
    "yolox_s_glfusion_640x640_bs8": ("gl", "gl_s_seed0", 640, 640, 8),
    # probes of the batch-size / batches-in-flight trade-off (tools/probe/subbatch.sh), not the metric's configuration
    "yolox_s_glfusion_1344x800_bs4": ("gl", "gl_s_seed0", 800, 1344, 4),
    "yolox_s_glfusion_1344x800_bs16": ("gl", "gl_s_seed0", 800, 1344, 16),
    "yolox_s_base_640x640_bs8": ("base", "base_s_seed0", 640, 640, 8),
    # UFPMP-Det detectors (configs/UFPMP-Det/*.py): ResNet-50 (+ GL-fusion plug-in) + FPN + GFLHead / MPHead
    "mp_det_res50_gl_1344x800_bs8": ("mpdet_gl", None, 800, 1344, 8),
    "mp_det_res50_1344x800_bs8": ("mpdet", None, 800, 1344, 8),
    "coarse_det_1344x800_bs8": ("gfl", None, 800, 1344, 8),
}
RESDET = ("gfl", "mpdet", "mpdet_gl")
DETECTOR_NAME = {"gl": "YOLOX-s + GL-fusion neck", "base": "YOLOX-s", "gfl": "GFL ResNet-50 + FPN",
                 "mpdet": "MPDet ResNet-50 + FPN + MPHead",
                 "mpdet_gl": "MPDet ResNet-50 + GL-fusion (x + Patch_Conv_NonLocal_new(x) on C3-C5) + FPN + MPHead"}
SECONDARY = "mp_det_res50_gl_1344x800_bs8"      # BASELINE config 3 as named


def resdet_algorithmic(kind, H, W, nc=10, proxies=42):
    """Algorithmic (GMAC conv, GMAC matmul, HBM bytes fp16) per image of ResNet-50 (+ GL plug-in) + FPN(start 1,
    5 outs, extra on_output) + GFLHead/MPHead by SURVEY 8d's rule: conv MACs = out elems x Cin x k^2
    with the REAL channel counts; bytes = each conv reads its input once and writes its output once;
    matmul MACs = the two batched products of every Non_local_Block as the reference evaluates them
    (N x C x N each, Identity_Conv.py:162-167)."""
    mac = [0.0]
    byt = [0.0]
    mm = [0.0]

    def conv(h, w, cin, cout, k, s=1, pad=None):
        pad = k // 2 if pad is None else pad
        ho, wo = (h + 2 * pad - k) // s + 1, (w + 2 * pad - k) // s + 1
        mac[0] += ho * wo * cout * cin * k * k
        byt[0] += 2.0 * (h * w * cin + ho * wo * cout)
        return ho, wo
    h, w = conv(H, W, 3, 64, 7, 2)
    h, w = (h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1
    cin, sizes = 64, []
    for i, nb in enumerate((3, 4, 6, 3)):
        planes = 64 * 2 ** i
        for j in range(nb):
            s = 2 if (j == 0 and i > 0) else 1
            conv(h, w, cin, planes, 1)
            ho, wo = conv(h, w, planes, planes, 3, s)
            if j == 0:
                conv(h, w, cin, planes * 4, 1, s, 0)
            conv(ho, wo, planes, planes * 4, 1)
            h, w, cin = ho, wo, planes * 4
        sizes.append((h, w, cin))
    if kind == "mpdet_gl":          # GL-fusion plug-in on C3..C5: Patch_Conv_NonLocal_new(C, C, channel_scale=1): four quadrant
        for (h, w, c) in sizes[1:]:  # non-local blocks with inter_channels = C (Non_local_family.py:208-228) + 1x1 channel_conv
            hh, hw = h // 2, w // 2
            for (qh, qw) in ((hh, hw), (h - hh, hw), (hh, w - hw), (h - hh, w - hw)):
                ci = c
                for _ in range(3):
                    conv(qh, qw, c, ci, 1)          # theta, phi, g
                conv(qh, qw, ci, c, 1)              # conv_out
                mm[0] += 2.0 * (qh * qw) * ci * (qh * qw)
            conv(h, w, c, c, 1)                     # channel_conv ('linear')
    lv = []
    for (h, w, c) in sizes[1:]:
        conv(h, w, c, 256, 1)
        conv(h, w, 256, 256, 3)
        lv.append((h, w))
    for _ in range(2):
        lv.append(conv(lv[-1][0], lv[-1][1], 256, 256, 3, 2))
    for (h, w) in lv:
        for _ in range(8):
            conv(h, w, 256, 256, 3)
        conv(h, w, 256, 68, 3)
        if kind == "gfl":
            conv(h, w, 256, nc, 3)
        else:
            conv(h, w, 256, 256, 3)
            mm[0] += h * w * 256 * proxies
    return mac[0] / 1e9, mm[0] / 1e9, byt[0]


def synthetic_state_dict(tag):
    """Seeded random-init weights of the named architecture.  BN running stats are the
    calibrated ones stored with the golden vectors (tests/golden): everything else is the
    pure function glsdet_amd.synth.synth_tensor(key, shape, seed)."""
    from glsdet_amd.synth import synth_state_dict
    kind, phi = tag.split("_")[0], tag.split("_")[1]
    with open(os.path.join(ROOT, "tests", "golden", "shapes.json")) as f:
        shapes = json.load(f)["%s_%s" % (kind, phi)]
    sd = synth_state_dict(shapes, 0)
    g = np.load(os.path.join(ROOT, "tests", "golden", "drone_golden.npz"))
    pre = "model/%s/bn/" % tag
    for k in g.files:
        if k.startswith(pre):
            sd[k[len(pre):]] = torch.from_numpy(g[k])
    return sd


def _host_threads():
    try:
        ncpu = len(os.sched_getaffinity(0))
    except AttributeError:
        ncpu = os.cpu_count() or 1
    return max(1, min(ncpu, 64))


def cpu_baseline(sd, kind, n, H, W, conf, nms_thr, budget_s=25.0):
    """The oracle (a port: kind='port') timed on the host cores: forward + decode + NMS."""
    from oracle import glsdet_oracle as O
    torch.set_num_threads(_host_threads())
    x = O.synth_input((n, 3, H, W), 100)

    def step():
        with torch.no_grad():
            outs = O.FORWARDS[kind](sd, x)
            dec = O.decode_outputs(outs, (H, W))
            O.non_max_suppression(dec, 10, (H, W), np.array([H, W]), False, conf, nms_thr)
    t0 = time.perf_counter()
    step()                                   # warm
    first = time.perf_counter() - t0
    iters = int(max(1, min(5, (budget_s - first) // max(first, 1e-3))))
    t0 = time.perf_counter()
    for _ in range(iters):
        step()
    dt = (time.perf_counter() - t0) / iters
    return {"value": round(n / dt, 3), "unit": "img/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%d timed iters (1 warm) of the same batch: %dx3x%dx%d fp32, forward+decode+NMS, torch CPU oracle"
                      % (iters, n, H, W)}


def resdet_cpu_baseline(sd, kind, H, W, thr, max_per_img, budget_s=25.0):
    """oracle/mpdet_oracle.py timed on the host cores on ONE image of the benchmark shape
    (a bs-8 batch of ResNet-50 at 800x1344 does not fit the time budget)."""
    from oracle import glsdet_oracle as O
    from oracle import mpdet_oracle as M
    from glsdet_amd.resdet import HipGflDetector
    torch.set_num_threads(_host_threads())
    x = O.synth_input((1, 3, H, W), 100)
    fwd = {"gfl": lambda: M.gfl_forward(sd, x),
           "mpdet": lambda: M.mpdet_forward(sd, x, HipGflDetector.DEFAULTS["proxies_list"]),
           "mpdet_gl": lambda: M.mpdet_forward(sd, x, HipGflDetector.DEFAULTS["proxies_list"], gl_fusion=True)}[kind]

    def step():
        with torch.no_grad():
            c, r = fwd()
            M.gfl_get_bboxes(c, r, [8, 16, 32, 64, 128], [(H, W, 3)], thr, 1000, 0.6, max_per_img)
    t0 = time.perf_counter()
    step()
    first = time.perf_counter() - t0
    iters = int(max(1, min(5, (budget_s - first) // max(first, 1e-3))))
    t0 = time.perf_counter()
    for _ in range(iters):
        step()
    dt = (time.perf_counter() - t0) / iters
    return {"value": round(1 / dt, 3), "unit": "img/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%d timed iters (1 warm) of ONE image 1x3x%dx%d fp32 (the bs-8 batch exceeds the time budget), "
                      "forward+decode+NMS, torch CPU restatement (oracle/mpdet_oracle.py)" % (iters, H, W)}


def calibrate_resdet(sd, kind, img, args, dev):
    """Random-init class predictors pass almost every (position, class) pair.  To load the
    post-processing like a real image (about --candidates pairs per image above the threshold):
    GFLHead: one common shift of the gfl_cls bias (threshold stays the config's 0.05);
    MPHead: the scores are cosines to random proxies (no bias to shift), so the score threshold
    itself is bisected and reported in config.score_thr.  Setup only, not timed."""
    from glsdet_amd.resdet import HipGflDetector
    cls, _ = HipGflDetector(kind, sd, dtype=args.dtype, device=dev).forward_raw(img)
    logits = torch.cat([c.flatten(1) for c in cls], 1)
    if kind == "gfl":
        lo, hi = -60.0, 20.0
        for _ in range(40):
            mid = 0.5 * (lo + hi)
            n = float((torch.sigmoid(logits + mid) > 0.05).sum(1).float().mean())
            lo, hi = (mid, hi) if n < args.candidates else (lo, mid)
        sd = dict(sd)
        sd["bbox_head.gfl_cls.bias"] = sd["bbox_head.gfl_cls.bias"] + 0.5 * (lo + hi)
        return sd, 0.05
    p = torch.sigmoid(logits)
    lo, hi = 0.0, 1.0
    for _ in range(40):
        mid = 0.5 * (lo + hi)
        n = float((p > mid).sum(1).float().mean())
        lo, hi = (mid, hi) if n > args.candidates else (lo, mid)
    return sd, round(0.5 * (lo + hi), 4)


def _torch_decode(outs, H, W):
    """decode_outputs (drone/models/core/utils_bbox.py:254-306) in torch on the device, for the
    calibration below only (data preparation; the measured path uses glsdet_yolox_decode)."""
    flat = torch.cat([o.flatten(2) for o in outs], 2).permute(0, 2, 1).clone()
    flat[..., 4:] = torch.sigmoid(flat[..., 4:])
    grids, strides = [], []
    for o in outs:
        h, w = o.shape[-2:]
        gy, gx = torch.meshgrid(torch.arange(h, device=o.device), torch.arange(w, device=o.device), indexing="ij")
        grids.append(torch.stack((gx, gy), 2).reshape(1, -1, 2).float())
        strides.append(torch.full((1, h * w, 1), H / h, device=o.device))
    grids, strides = torch.cat(grids, 1), torch.cat(strides, 1)
    flat[..., :2] = (flat[..., :2] + grids) * strides
    flat[..., 2:4] = torch.exp(flat[..., 2:4]) * strides
    flat[..., [0, 2]] = flat[..., [0, 2]] / W
    flat[..., [1, 3]] = flat[..., [1, 3]] / H
    return flat.contiguous()


def calibrate_yolox_head(sd, kind, img, args, dev):
    """Make the random-init head behave like a trained one for the post-processing load (weights only; the
    timed path is untouched; SURVEY 8d "dense mode"):
      * objectness: one common bias shift so that about --candidates anchors per image pass --conf;
      * box branch: random reg_preds saturate exp(w), exp(h) (infinite boxes, nothing overlaps, the NMS fixed
        point converges in one sweep).  The reg_preds weights are rescaled per level so that the xy offsets have
        std 0.5 cell and log(w/stride), log(h/stride) ~ N(mu, 0.35^2); mu is bisected until the class-wise NMS
        at --nms suppresses about --suppress of the candidates (a trained detector's clustered duplicates).
    The suppression is evaluated with the product's own NMS kernel on torch-decoded boxes."""
    from glsdet_amd.detector import HipDetector
    from glsdet_amd.engine import Engine
    outs = HipDetector(kind, sd, dtype=args.dtype, device=dev).forward_raw(img)
    obj = torch.cat([o[:, 4].flatten(1) for o in outs], 1)
    cls = torch.cat([torch.sigmoid(o[:, 5:]).max(1)[0].flatten(1) for o in outs], 1)
    lo, hi = -80.0, 20.0
    for _ in range(40):
        mid = 0.5 * (lo + hi)
        n = float((torch.sigmoid(obj + mid) * cls >= args.conf).sum(1).float().mean())
        lo, hi = (mid, hi) if n < args.candidates else (lo, mid)
    shift = 0.5 * (lo + hi)
    sd = dict(sd)
    scales = []
    for k, o in enumerate(outs):
        b = sd["head.reg_preds.%d.bias" % k].to(o.device).view(1, 4, 1, 1)
        r = o[:, :4] - b
        scales.append((0.5 / float(r[:, :2].std().clamp(min=1e-6)), 0.35 / float(r[:, 2:].std().clamp(min=1e-6))))
        sd["head.obj_preds.%d.bias" % k] = sd["head.obj_preds.%d.bias" % k] + shift
    eng = Engine("f32", dev)
    n, A = outs[0].shape[0], sum(o.shape[2] * o.shape[3] for o in outs)
    nb = eng.nms_buffers(n, A, A, args.max_det)

    def suppressed(mu):
        mod = []
        for k, o in enumerate(outs):
            b = sd["head.reg_preds.%d.bias" % k].to(o.device).view(1, 4, 1, 1)
            t = o.clone()
            t[:, :2] = (o[:, :2] - b[:, :2]) * scales[k][0]
            t[:, 2:4] = (o[:, 2:4] - b[:, 2:]) * scales[k][1] + mu
            t[:, 4] = o[:, 4] + shift
            mod.append(t)
        dets, count, status = eng.nms(_torch_decode(mod, img.shape[2], img.shape[3]), 10, 0, args.conf, args.nms, nb)
        torch.cuda.synchronize()
        cand = nb["ws"][: 4 * n].view(torch.int32).float().sum()
        return 1.0 - float(count[n:2 * n].float().sum() / cand.clamp(min=1))
    lo, hi = -1.0, 4.5
    for _ in range(14):
        mid = 0.5 * (lo + hi)
        lo, hi = (mid, hi) if suppressed(mid) < args.suppress else (lo, mid)
    mu = 0.5 * (lo + hi)
    for k in range(len(outs)):
        w = sd["head.reg_preds.%d.weight" % k].clone()
        w[:2] *= scales[k][0]
        w[2:] *= scales[k][1]
        sd["head.reg_preds.%d.weight" % k] = w
        sd["head.reg_preds.%d.bias" % k] = torch.tensor([0.0, 0.0, mu, mu])
    return sd, {"objectness_bias_shift": round(shift, 3), "log_box_over_stride_mean": round(mu, 3)}


# --------------------------------------------------------------------------------------------- one workload
def run_workload(args, workload, rank, world, dev, with_cpu_baseline, nsteps=None, nwindows=None, leg=""):
    """Build, calibrate, time (warmup + exactly --steps steps between barriers) and profile per op ONE workload.
    -> (result dict, elapsed seconds) on every rank."""
    import torch.distributed as dist
    from glsdet_amd.dist import DetectionExchange, ranks_agree

    nsteps = args.steps if nsteps is None else nsteps
    nwindows = args.windows if nwindows is None else nwindows
    kind, tag, H, W, bs = WORKLOADS[workload]
    gen = torch.Generator(device=dev).manual_seed(rank)
    img = torch.randn(bs, 3, H, W, generator=gen, device=dev)
    nstreams = 1 if args.no_graph else max(1, args.streams)
    calib = {}
    if kind in RESDET:
        from glsdet_amd.resdet import HipGflDetector
        from glsdet_amd.synth import synth_input, synth_resdet_state_dict
        extra = dict(gl_fusion=True) if kind == "mpdet_gl" else {}
        sd = synth_resdet_state_dict("mpdet" if kind == "mpdet_gl" else kind, 0, synth_input((1, 3, 128, 160), 100), **extra)
        hip_kind = "mpdet" if kind == "mpdet_gl" else kind
        sd, score_thr = calibrate_resdet(sd, hip_kind, img, args, dev)
        det = HipGflDetector(hip_kind, sd, dtype=args.dtype, device=dev, autotune=not args.no_autotune)
        post = dict(score_thr=score_thr, iou_thr=0.6, nms_pre=1000, max_per_img=100 if kind == "gfl" else 500,
                    max_cand=16384)
        buf = lambda ci: ci.nb
    else:
        from glsdet_amd.detector import HipDetector
        sd = synthetic_state_dict(tag)
        sd, calib = calibrate_yolox_head(sd, kind, img, args, dev)          # setup only, not timed
        det = HipDetector(kind, sd, dtype=args.dtype, device=dev, autotune=not args.no_autotune)
        post = dict(conf_thres=args.conf, nms_thres=args.nms, max_det=args.max_det)
        buf = lambda ci: ci.nmsb
    if world > 1:
        full = post.get("max_det", post.get("max_per_img", 1000))
        post["exchange_cap"] = full if args.exchange_cap <= 0 else min(args.exchange_cap, full)
        if not args.no_autotune:
            # rank 0 measures the kernel variants, every rank runs ITS table: tuned independently, near-ties fall differently
            # per GPU and the slowest rank's choice would set the weak-scaling step (glsdet_amd.dist.share_tuning)
            from glsdet_amd.dist import share_tuning
            if rank == 0:
                det.compile(bs, H, W, post, use_graph=not args.no_graph, instance=0)
            calib["tuning_table_entries_shared_from_rank0"] = share_tuning(0)
    cs = [det.compile(bs, H, W, post, use_graph=not args.no_graph, instance=i) for i in range(nstreams)]
    for ci in cs:
        ci.img.copy_(img)                                        # resident in HBM before timing
    c = cs[0]
    exch = [DetectionExchange(buf(ci)["packed"]) for ci in cs] if world > 1 else None
    # results leave the device every step: the per-image counts of a batch are copied to pinned host memory on the
    # batch's own stream and consumed (summed) when that plan instance comes round again
    h_count = [torch.zeros(2 * bs, dtype=torch.int32).pin_memory() for _ in cs]
    ev = [torch.cuda.Event() for _ in cs]
    pending = [False] * len(cs)
    host_img = img.cpu().pin_memory() if args.from_host else None
    torch.cuda.synchronize()
    turn = [0]
    read_back = [0, 0]                     # detections, batches whose counts reached the host

    def consume(i):
        if pending[i]:
            ev[i].synchronize()
            read_back[0] += int(h_count[i][:bs].sum())
            read_back[1] += 1
            pending[i] = False

    def step():
        i = turn[0] % nstreams
        ci = cs[i]
        turn[0] += 1
        consume(i)
        st = ci.graph_stream if not args.no_graph else torch.cuda.current_stream()
        if args.no_graph:
            det.run(ci)
        else:
            # one batch = one graph replay on the instance's own stream (+ its gather when N>1);
            # consecutive batches rotate over the instances, so `--streams` batches are in flight
            if host_img is not None:
                with torch.cuda.stream(st):
                    ci.img.copy_(host_img, non_blocking=True)
            det.run_async(ci)
        with torch.cuda.stream(st):
            if world > 1:
                exch[i].gather()
            h_count[i].copy_(buf(ci)["count"], non_blocking=True)
            ev[i].record(st)
        pending[i] = True

    def fence():
        for i in range(len(cs)):
            consume(i)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # settle (untimed, before the W warmup steps): the first GPU process on a fresh box ran the same build at 3000-3200
    # instead of 4100-4300 img/s for its first ~0.2 s twice this round (clocks / power state still ramping); a run of the
    # real step until --settle seconds have passed puts the measurement on the steady state the later steps see anyway
    # (N > 1: every round carries collectives, so the ranks agree on each further round -- a clock of its own per rank would
    # let one rank leave the loop a round early and pair its warmup gathers with the others' barrier)
    t_settle = time.perf_counter()
    while ranks_agree(time.perf_counter() - t_settle < args.settle, "any", dev):
        for _ in range(8):
            step()
        fence()
    for _ in range(args.warmup):
        step()
    fence()
    # R windows of EXACTLY --steps steps, each bracketed by barrier + synchronize on both sides and maxed over the ranks; the
    # line's value is the MEDIAN window (a single 37 ms window used to decide the headline under the driver's --steps 20)
    windows = []
    for _ in range(max(1, nwindows)):
        read_back[0] = read_back[1] = 0
        t0 = time.perf_counter()
        for _ in range(nsteps):
            step()
        fence()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        assert read_back[1] == nsteps, "every timed batch must have delivered its counts to the host"
        windows.append(el)
    elapsed = float(np.median(windows))
    dets = det.collect(c)                    # also checks the NMS capacity/overflow flags
    if kind in RESDET:
        dets = [d[0] for d in dets]

    # ---- per-op timing of the same plan (eager replay, HIP events on the launch stream)
    ops = c.plan.ops()
    st = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(st):
        c.plan.run_timed(st, reps=2)         # warm
        ms = c.plan.run_timed(st, reps=10)
    torch.cuda.synchronize()
    conv = [(o, t) for o, t in zip(ops, ms) if o["kind"] == 0]
    executed_conv_flops = sum(o["flops"] for o, _ in conv)
    if kind in RESDET:
        alg_conv_gmac, alg_mm_gmac, alg_bytes = resdet_algorithmic(kind, H, W)
    else:
        alg_conv_gmac, alg_mm_gmac = ALGORITHMIC_GMAC[workload]
        alg_bytes = 431.8e6 if (H, W) == (800, 1344) else 164.5e6
    conv_flops = 2.0 * alg_conv_gmac * 1e9 * bs
    conv_ms = sum(t for _, t in conv)
    all_ms = float(ms.sum())
    peak = PEAK_F16_TFLOPS if args.dtype == "f16" else PEAK_F32_TFLOPS
    achieved = conv_flops / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
    wall_ms = elapsed / nsteps * 1e3
    e2e = 2.0 * (alg_conv_gmac + alg_mm_gmac) * 1e9 * bs / (wall_ms * 1e-3) / 1e12
    # the same two figures on the FLOPs the launched kernels execute (weight composition / re-association remove algorithmic
    # work: `frac` then says how fast the JOB is, `frac_executed` how busy the MFMA pipe is)
    executed_all_flops = sum(o["flops"] for o in ops)
    achieved_x = executed_conv_flops / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
    e2e_x = executed_all_flops / (wall_ms * 1e-3) / 1e12
    # HBM traffic per conv launch: PMC counters cannot be read from inside the benchmark; the
    # latest committed rocprofv3 --pmc summary of this same command (tools/profile_round.sh ->
    # profiles/*/traffic.json, FETCH_SIZE doubled per the gfx950 correction) is reported.
    traffic = None
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "traffic.json")))      # round tags sort by name
    for cand in reversed(cands):
        with open(cand) as f:
            tj = json.load(f)
        if tj.get("workload", "yolox_s_glfusion_1344x800_bs8") == workload and args.dtype == "f16":
            traffic = round(tj["hbm_bytes_per_launch"])
            break
    roofline = {"bound": "mfma", "kernel": "conv family: conv_igemm + conv_halo + conv1x1_ws + fused bottleneck (all instantiations)",
                "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
                "achieved_executed": round(achieved_x, 2), "frac_executed": round(achieved_x / peak, 4),
                "achieved_end_to_end": round(e2e, 2), "frac_end_to_end": round(e2e / peak, 4),
                "frac_end_to_end_executed": round(e2e_x / peak, 4),
                "algorithmic_over_executed_conv_flops": round(conv_flops / max(executed_conv_flops, 1.0), 4),
                "frac_note": "frac: algorithmic conv FLOPs / sum of conv launch durations, ONE stream (HIP events); "
                             "frac_end_to_end: all algorithmic FLOPs / wall clock per step, %d batches in flight; *_executed: the same "
                             "over the FLOPs the launched kernels execute (sum of the plan's op records)" % nstreams,
                "traffic": traffic, "traffic_unit": "HBM bytes per conv launch (rocprofv3 PMC, profiles/)",
                "algorithmic_bytes_per_launch": round(bs * alg_bytes / max(1, len(conv))),
                "launches_per_step": len(conv),
                "avg_launch_us": round(conv_ms * 1e3 / max(1, len(conv)), 2),
                "gflop_per_launch": round(conv_flops / max(1, len(conv)) / 1e9, 3),
                "conv_ms_per_step": round(conv_ms, 4), "all_ops_ms_per_step_eager": round(all_ms, 4),
                "algorithmic_gflop_per_image": round(2.0 * (alg_conv_gmac + alg_mm_gmac), 2),
                "executed_conv_gflop_per_image": round(executed_conv_flops / bs / 1e9, 2)}
    if args.op_table and rank == 0:
        path = args.op_table if (workload == args.workload and not leg) else args.op_table + "." + (leg or workload)
        with open(path, "w") as f:
            f.write("idx\tkind\tms\tgflop\ttflops\tMB\tGBps\tname\n")
            for i, (o, t) in enumerate(zip(ops, ms)):
                f.write("%d\t%d\t%.4f\t%.3f\t%.1f\t%.2f\t%.0f\t%s\n" % (
                    i, o["kind"], t, o["flops"] / 1e9, o["flops"] / max(t, 1e-6) / 1e9, o["bytes"] / 1e6,
                    o["bytes"] / max(t, 1e-6) / 1e6, o["name"]))

    if kind in RESDET:      # stage-1 counters: [image][level] pairs above the threshold; NMS input = per-level top-k merged
        nl = len(c.cls)
        cand_counts = c.nb["ws"][: 4 * bs * nl].view(torch.int32).view(bs, nl).sum(1).cpu().tolist()
        nms_in = [int(min(v, post["nms_pre"] * nl)) for v in cand_counts]
        unclamped = c.nb["count"][bs:2 * bs].cpu().tolist()
    else:
        cand_counts = [int(v) for v in c.nmsb["ws"][: 4 * bs].view(torch.int32).cpu().tolist()]
        nms_in = cand_counts
        unclamped = c.nmsb["count"][bs:2 * bs].cpu().tolist()
    suppressed = 1.0 - float(sum(unclamped)) / max(1, sum(nms_in))
    n_img = bs * world * nsteps
    res = {"workload": workload, "value": round(n_img / elapsed, 2), "unit": "img/s",
           "ms_per_step": round(wall_ms, 4), "steps": nsteps,
           "windows": {"n": len(windows), "steps_each": nsteps, "statistic": "median",
                       "ms_per_step_min": round(min(windows) / nsteps * 1e3, 4), "ms_per_step_max": round(max(windows) / nsteps * 1e3, 4)},
           "config": {"workload": workload, "detector": DETECTOR_NAME[kind],
                      "input": [bs, 3, H, W], "images_per_gpu": bs, "global_batch": bs * world, "num_classes": 10,
                      "post": post, "head_calibration": calib, "hip_graph": not args.no_graph,
                      "batches_in_flight": nstreams, "settle_seconds_untimed": args.settle,
                      "input_from_host_each_step": bool(args.from_host),
                      "detections_per_image_rank0": [int(len(d)) for d in dets],
                      "candidates_per_image_rank0": cand_counts,
                      "suppressed_frac": round(suppressed, 4),
                      "detections_read_back_in_timed_loop_rank0": read_back[0],
                      "parallelism": "image-sharded dp%d, one all_gather of detections per step" % world,
                      "exchange": ({"rows_per_image": post["exchange_cap"],
                                    "images_truncated_by_the_cap_rank0": int(sum(1 for v in unclamped if v > post["exchange_cap"]))}
                                   if world > 1 else None)},
           "roofline": roofline}
    if with_cpu_baseline:
        res["cpu_baseline"] = resdet_cpu_baseline(sd, kind, H, W, post["score_thr"], post["max_per_img"]) \
            if kind in RESDET else cpu_baseline(sd, kind, bs, H, W, args.conf, args.nms)
    del cs, c, det, exch
    torch.cuda.empty_cache()
    return res


def run_two_stage(args, rank, world, dev):
    """BASELINE configs[4]: the two-stage UFPMP-Det evaluation flow (ufpmp_det_eval.py:253-300) -- coarse GFL detector,
    host packing, device mosaic, fine MPDet, device back-mapping + merge NMS -- with the two detectors pipelined on separate
    HIP streams (glsdet_amd.ufp.TwoStagePipeline, three lanes).  Synthetic 540 x 1024 BGR frames (VisDrone's common size;
    the data set is not available), synthetic weights, thresholds bisected so that the coarse stage proposes ~80 boxes and
    the fine stage ~600 candidates per mosaic.  Frames shard round-robin over the ranks; per pass ONE all_gather of the
    per-frame detection counts + the first `cap` boxes (N > 1).  A "step" here is one pass over the rank's 32 frames."""
    import torch.distributed as dist
    from glsdet_amd.dist import ranks_agree
    from glsdet_amd.resdet import HipGflDetector
    from glsdet_amd.synth import synth_input, synth_resdet_state_dict
    from glsdet_amd.ufp import TwoStagePipeline, UfpSecondStage, two_stage_detect

    def frame(seed, h=540, w=1024):
        rng = np.random.default_rng([seed, 0x1A6E])
        yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
        img = np.stack([127 + 120 * np.sin(xx / (5 + 3 * c) + seed) * np.cos(yy / (7 + 2 * c)) for c in range(3)], -1)
        img += rng.normal(0, 25, img.shape)
        img[: h // 5, : w // 4] = 255
        img[-(h // 6):, -(w // 5):] = 0
        return np.clip(img, 0, 255).astype(np.uint8)[:, :, ::-1].copy()
    def agree(ok):
        """Every rank learns whether all ranks are fine, so that a failure on one rank ends this leg on all of them before
        the next collective instead of leaving the others waiting in it (the headline line must still be printed)."""
        return ranks_agree(ok, "all", dev)

    def failed(where, err):
        return {"workload": "ufpmp_two_stage_540x1024", "n_gpus": world,
                "error": "%s: %s" % (where, err or "another rank failed (this one was fine)")}
    per_rank, lanes, cap = 32, 3, 500
    S, err = {}, None

    def setup():
        calib = synth_input((1, 3, 128, 160), 100)
        coarse = HipGflDetector("gfl", synth_resdet_state_dict("gfl", 0, calib), dtype=args.dtype, autotune=not args.no_autotune)
        fine = HipGflDetector("mpdet", synth_resdet_state_dict("mpdet", 1, calib), dtype=args.dtype, autotune=not args.no_autotune)
        stage = UfpSecondStage()

        def thr_for(det, x, keep):
            cls, _ = det.forward_raw(x)
            p = torch.sigmoid(torch.cat([c.flatten() for c in cls]))
            return float(torch.topk(p, keep).values[-1])
        img0 = frame(2)
        x1, _ = stage.pipeline_input(torch.from_numpy(img0).to(dev).contiguous())
        c1 = dict(score_thr=thr_for(coarse, x1, 80), iou_thr=0.6, nms_pre=1000, max_per_img=100)
        _, mid = two_stage_detect(coarse, fine, img0, stage, c1, dict(score_thr=0.9999, iou_thr=0.6))
        x2, _ = stage.pipeline_input(mid["canvas"])
        c2 = dict(score_thr=thr_for(fine, x2, 600), iou_thr=0.6, nms_pre=1000, max_per_img=500)
        S["frames"] = [frame(100 + (rank + world * i) % 8) for i in range(per_rank)]          # global frame g = rank + world * i
        S["pipe"] = TwoStagePipeline(coarse, fine, stage, c1, c2, workers=lanes)
        S["pipe"].run(S["frames"][:8] * lanes)                    # every lane compiles (and tunes) every mosaic shape once
        S["record"] = torch.zeros(per_rank, cap + 1, 6, dtype=torch.float32, device=dev)
        S["gathered"] = torch.empty((world * per_rank, cap + 1, 6), dtype=torch.float32, device=dev) if world > 1 else None
        S.update(c1=c1, c2=c2, mid=mid)

    def local_pass():
        if os.environ.get("GLSDET_BENCH_TERTIARY_FAIL", "") == str(rank):       # rehearsal of the failure path below
            raise RuntimeError("injected failure on rank %d" % rank)
        res = S["pipe"].run(S["frames"])
        host = np.zeros((per_rank, cap + 1, 6), np.float32)
        for i, per_class in enumerate(res):
            rows = [np.concatenate([a[:, :5], np.full((len(a), 1), c, np.float32)], 1) for c, a in enumerate(per_class) if len(a)]
            rows = np.concatenate(rows)[:cap] if rows else np.zeros((0, 6), np.float32)
            host[i, : len(rows)] = rows
            host[i, cap, 0] = len(rows)
        S["record"].copy_(torch.from_numpy(host))
        return res

    def one_pass():
        """The rank's 32 frames, then (N > 1) the agreement and the ONE all_gather.  None = some rank failed."""
        nonlocal err
        res = None
        try:
            res = local_pass()
        except Exception as e:                                    # reported in the line, see failed()
            err = repr(e)
        if not agree(err is None):
            return None
        if world > 1:
            dist.all_gather_into_tensor(S["gathered"], S["record"])
        return res
    try:
        setup()
    except Exception as e:
        err = repr(e)
    if not agree(err is None):
        return failed("setup", err)
    c1, c2, mid = S["c1"], S["c2"], S["mid"]
    if one_pass() is None:
        return failed("first pass", err)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    passes = max(5, args.steps // 40)          # >= 160 frames per rank: the two-pass figure of round 2 moved 2 % run to run
    t0 = time.perf_counter()
    for _ in range(passes):
        res = one_pass()
        if res is None:
            return failed("timed pass", err)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    n_det = [int(sum(len(a) for a in r)) for r in res]
    return {"workload": "ufpmp_two_stage_540x1024", "metric": "frames/sec, two-stage UFPMP-Det (coarse GFL -> packing -> mosaic -> fine MPDet -> merge NMS)",
            "value": round(world * per_rank * passes / elapsed, 1), "unit": "frames/s", "ms_per_frame": round(elapsed / (per_rank * passes) * 1e3, 3),
            "n_gpus": world, "passes": passes, "frames_per_rank_and_pass": per_rank, "data": "synthetic", "dtype": args.dtype,
            "config": {"frame": [540, 1024], "lanes": lanes, "coarse": "GFL ResNet-50 + FPN (configs/UFPMP-Det/coarse_det.py)",
                       "fine": "MPDet ResNet-50 + FPN + MPHead (configs/UFPMP-Det/mp_det_res50.py)",
                       "coarse_post": {k: (round(v, 4) if isinstance(v, float) else v) for k, v in c1.items()},
                       "fine_post": {k: (round(v, 4) if isinstance(v, float) else v) for k, v in c2.items()},
                       "chips_frame0": int(len(mid["chips"])), "mosaic_frame0": [int(mid["canvas"].shape[0]), int(mid["canvas"].shape[1])],
                       "detections_per_frame_rank0": n_det[:8], "exchange_cap": cap,
                       "parallelism": "frame-sharded dp%d, one all_gather of the per-frame results per pass" % world},
            "note": "BASELINE.json configs[4] (two-stage UFPMP eval, detectors pipelined on separate HIP streams) on synthetic frames: "
                    "VisDrone-val is not available offline"}


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N fresh processes (this one has not touched the GPU),
    rank r on GPU r, rendezvous on 127.0.0.1.  Rank 0 inherits stdout and prints the line."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    try:
        while any(p.poll() is None for p in procs):
            for p in procs:
                if p.poll() not in (None, 0):            # one rank failed: stop the others (exact PIDs)
                    rc = p.returncode
                    for q in procs:
                        if q.poll() is None:
                            q.terminate()
            time.sleep(0.2)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc or max(abs(p.returncode or 0) for p in procs)


def _stub_rank(spec):
    """CPU rehearsal of the launcher (tests/test_surface_cpu.py; GLSDET_BENCH_STUB="ok" | "die:<rank>:<code>"): the ranks meet
    over gloo as the real ones do over RCCL, then either all leave cleanly (rank 0 prints a line) or one rank dies with
    <code> while the others sit in a collective it will never join -- spawn_ranks must end them and return non-zero."""
    import torch.distributed as dist
    rank = int(os.environ["RANK"])
    dist.init_process_group("gloo")
    dist.barrier()
    if spec.startswith("die:"):
        _, r, code = spec.split(":")
        if rank == int(r):
            os._exit(int(code))
        dist.barrier()                                   # never completes: the launcher has to stop this rank
        time.sleep(600)
        return 0
    dist.barrier()
    if rank == 0:
        print(json.dumps({"stub": "ok", "n_gpus": int(os.environ["WORLD_SIZE"])}))
    dist.destroy_process_group()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="yolox_s_glfusion_1344x800_bs8", choices=sorted(WORKLOADS))
    ap.add_argument("--dtype", default="f16", choices=["f16", "f32"])
    ap.add_argument("--conf", type=float, default=0.25)
    ap.add_argument("--candidates", type=int, default=2000,
                    help="calibrate the objectness bias so that about this many anchors per image pass --conf")
    ap.add_argument("--suppress", type=float, default=0.5,
                    help="calibrate the box branch so that NMS suppresses about this fraction of the candidates")
    ap.add_argument("--max-det", type=int, default=3000)
    ap.add_argument("--exchange-cap", type=int, default=0,
                    help="rows per image of the all_gather record (N>1); 0 = the detector's own cap (max_det / max_per_img): "
                         "nothing a rank keeps is cut by the exchange")
    ap.add_argument("--windows", type=int, default=9, help="timed windows of exactly --steps steps; the median is reported")
    ap.add_argument("--no-legs", action="store_true", help="skip the extra legs (f32_strict, BASELINE config 2 at 640x640)")
    ap.add_argument("--nms", type=float, default=0.65)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--from-host", action="store_true",
                    help="diagnostic: every step first copies its fp32 batch from pinned host memory (the "
                         "PCIe-inclusive rate quoted in DESIGN.md; never the headline value)")
    ap.add_argument("--streams", type=int, default=3,
                    help="plan instances replayed round-robin on their own HIP streams (batches in flight)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend; gloo is for rehearsing the N>1 control flow on a box with fewer "
                         "GPUs than ranks (ranks then share devices round-robin) -- never for a reported number")
    ap.add_argument("--settle", type=float, default=1.5,
                    help="seconds of untimed steps before the warmup (clock / power ramp of a cold box); 0 = none")
    ap.add_argument("--no-autotune", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the BASELINE config-3 leg (`secondary`)")
    ap.add_argument("--no-tertiary", action="store_true", help="skip the BASELINE config-5 leg (`tertiary`: two-stage UFPMP pipeline)")
    ap.add_argument("--op-table", default="", help="write the per-op timing table (tsv) here")
    ap.add_argument("--hang-dump", type=int, default=0, metavar="SECONDS",
                    help="dump every thread's stack to stderr and exit if the run is still going after SECONDS (debugging aid)")
    args = ap.parse_args()
    if args.hang_dump and ("WORLD_SIZE" in os.environ or args.gpus == 1):        # the ranks, not the launcher
        import faulthandler
        faulthandler.dump_traceback_later(args.hang_dump, exit=True)

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus))                 # before any GPU call of this process
    if os.environ.get("GLSDET_BENCH_STUB") and "WORLD_SIZE" in os.environ:
        sys.exit(_stub_rank(os.environ["GLSDET_BENCH_STUB"]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d (launch one rank per GPU, or drop the launcher and let "
                 "bench.py start the ranks itself)" % (args.gpus, world))
    if args.backend == "gloo":
        local_rank %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    dev = "cuda:%d" % local_rank
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "gloo":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device(dev))

    def progress(what):
        if rank == 0:
            print("[bench %6.1fs] %s" % (time.perf_counter() - t_start, what), file=sys.stderr, flush=True)
    t_start = time.perf_counter()
    base = world == 1 and not args.no_cpu_baseline
    progress("workload %s on %d rank(s)" % (args.workload, world))
    main_res = run_workload(args, args.workload, rank, world, dev, base)
    second = None
    if not args.no_secondary and args.workload != SECONDARY:
        progress("secondary workload %s" % SECONDARY)
        second = run_workload(args, SECONDARY, rank, world, dev, base)
    legs = {}
    if not args.no_legs and not args.no_secondary and args.workload == "yolox_s_glfusion_1344x800_bs8" and args.dtype == "f16":
        import copy
        progress("leg f32_strict (the mode that meets north_star's 1e-4 / 1e-3: exact-f32 MFMA)")
        a32 = copy.copy(args)
        a32.dtype = "f32"
        r32 = run_workload(a32, args.workload, rank, world, dev, False, nsteps=max(5, min(args.steps, 40)), nwindows=3, leg="f32_strict")
        legs["f32_strict"] = dict(r32, dtype="f32", n_gpus=world,
                                  note="same workload, same protocol, fp32 storage + exact-f32 MFMA (v_mfma_f32_32x32x2_f32): the mode the "
                                       "f32 parity tests hold to max(1e-4, 2 x fp64 noise); roofline against the 157.3 TFLOP/s fp32 matrix peak")
        progress("leg config2 (BASELINE configs[1]: YOLOX-s + GL-fusion neck, 640x640 bs 8 fp16)")
        legs["config2"] = dict(run_workload(args, "yolox_s_glfusion_640x640_bs8", rank, world, dev, False, nwindows=min(args.windows, 5)),
                               dtype=args.dtype, n_gpus=world, note="BASELINE.json configs[1] at its named shape")
    third = None
    if not args.no_tertiary and not args.no_secondary and args.workload == "yolox_s_glfusion_1344x800_bs8" and not args.no_graph:
        progress("tertiary workload ufpmp_two_stage_540x1024")
        third = run_two_stage(args, rank, world, dev)
    progress("done")
    if rank == 0:
        line = {
            "metric": "images/sec fwd @1333x800 bs=8 (detection forward incl. decode+NMS)",
            "value": main_res["value"], "unit": "img/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": main_res["ms_per_step"], "windows": main_res["windows"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": main_res["config"], "roofline": main_res["roofline"],
        }
        if "cpu_baseline" in main_res:
            line["cpu_baseline"] = main_res["cpu_baseline"]
        if second is not None:
            line["secondary"] = dict(second, metric=line["metric"], n_gpus=world, steps=args.steps, warmup=args.warmup,
                                     dtype=args.dtype, note="BASELINE.json configs[2] (mp_det_res50: ResNet-50 + GL-fusion + "
                                     "decoupled MPHead, 1333x800 bs=8), same protocol as the primary line")
        if third is not None:
            line["tertiary"] = third
        for k, v in legs.items():
            line[k] = v
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
