"""torch custom ops over the C ABI: `torch.ops.glsdet.*` (north_star: "the hot path ... exposed through torch
custom ops"; SURVEY section 8b, C-ABI / custom-op layer).

Each op is one C-ABI call of libglsdet_hip.so on the CURRENT HIP stream of the input's device: inputs are borrowed
(they must be contiguous in the declared layout, else RuntimeError -- the reference's assertion style), outputs are
allocated by the op on the input's device, nothing synchronises.  There is no CPU implementation: a CPU tensor raises.

    glsdet::conv_bn_act(x, w, scale, bias, cout, R, S, stride, pad, act, res=None, out_fp32=False) -> y
        x NHWC [N,H,W,C] (C % 8 == 0) fp16 or fp32; w = pack_conv_weight(...) ; scale / bias fp32 [cout_pad];
        y NHWC [N,Ho,Wo,ceil8(cout)];  act: 0 none 1 silu 2 relu 3 lrelu 4 gelu 5 sigmoid (| 0x100 residual first)
        reference: BaseConv drone/models/base/baseConv.py:6-19 (+ Bottleneck add, darknet.py:61-62)
    glsdet::nonlocal_dot(x, tpg, ci, wout, bout) -> y
        Non_local_Block without its three projections: drone/models/block/non_local/Identity_Conv.py:157-173
    glsdet::yolox_decode(levels, num_classes, in_h, in_w, mode) -> pred [N, A, 5+nc]
        decode_outputs, drone/models/core/utils_bbox.py:254-306 (mode 1: mmdet YOLOXHead._bbox_decode)
    glsdet::nms(pred, num_classes, box_mode, conf_thres, nms_thres, max_det) -> (dets [N,max_det,7], count [2N], status [1])
        class max + threshold + per-class NMS of non_max_suppression, utils_bbox.py:375-419
    glsdet::batched_nms(boxes, scores, idxs, iou_threshold) -> keep int64 [K]
        torchvision.ops.boxes.batched_nms's signature (utils_bbox.py:414-419), one image

Importing this module registers the ops (idempotent).  `pack_conv_weight` / `fold_bn` are host-side helpers.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import F16, F32, ConvDesc, View, check

_DT = {torch.float16: F16, torch.float32: F32}


def _stream(t: torch.Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream


def _need_cuda(*ts):
    for t in ts:
        if t is not None and t.device.type != "cuda":
            raise RuntimeError("glsdet ops run on an MI355X only (got a %s tensor): there is no CPU fallback" % t.device.type)


def _nhwc_view(t: torch.Tensor, what: str) -> View:
    if t.dim() != 4 or not t.is_contiguous() or t.dtype not in _DT:
        raise RuntimeError("%s must be a contiguous NHWC [N,H,W,C] fp16 / fp32 tensor" % what)
    n, h, w, c = t.shape
    if c % 8:
        raise RuntimeError("%s: channels must be a multiple of 8 (got %d)" % (what, c))
    st = t.untyped_storage()
    return View(t.data_ptr(), h * w * c, w * c, c, n, h, w, c, _DT[t.dtype], 0, st.data_ptr(), st.data_ptr() + st.nbytes())


def ceil_to(v: int, m: int) -> int:
    return (v + m - 1) // m * m


def pack_conv_weight(w: torch.Tensor, scale: torch.Tensor, bias: torch.Tensor, cin_pad: int, dtype: torch.dtype,
                     device="cuda") -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """OIHW weights + per-channel scale / bias -> (packed [cout_pad, kpad] in `dtype`, scale fp32 [cout_pad], bias fp32
    [cout_pad]) as glsdet_conv2d wants them: k = (r*S + s)*cin_pad + ci, rows / K zero padded."""
    lib = _lib.load()
    cout, cin, R, S = w.shape
    assert cin <= cin_pad and cin_pad % 8 == 0
    kpad = lib.glsdet_conv_kpad(R, S, cin_pad, _DT[dtype])
    cpad = lib.glsdet_conv_cout_pad(ceil_to(cout, 8))
    wp = torch.zeros(cpad, R, S, cin_pad, dtype=torch.float32)
    wp[:cout, :, :, :cin] = w.detach().float().cpu().permute(0, 2, 3, 1)
    flat = torch.zeros(cpad, kpad, dtype=torch.float32)
    flat[:, : R * S * cin_pad] = wp.reshape(cpad, -1)
    sc, bi = torch.ones(cpad), torch.zeros(cpad)
    sc[:cout], bi[:cout] = scale.detach().float().cpu(), bias.detach().float().cpu()
    return flat.to(dtype).to(device), sc.to(device), bi.to(device)


_REGISTERED = False


def register():
    global _REGISTERED
    if _REGISTERED:
        return
    _REGISTERED = True
    lib_def = torch.library.Library("glsdet", "DEF")
    lib_def.define("conv_bn_act(Tensor x, Tensor w, Tensor scale, Tensor bias, int cout, int R, int S, int stride, int pad, "
                   "int act, Tensor? res=None, bool out_fp32=False) -> Tensor")
    lib_def.define("nonlocal_dot(Tensor x, Tensor tpg, int ci, Tensor wout, Tensor bout) -> Tensor")
    lib_def.define("yolox_decode(Tensor[] levels, int num_classes, int in_h, int in_w, int mode=0) -> Tensor")
    lib_def.define("nms(Tensor pred, int num_classes, int box_mode, float conf_thres, float nms_thres, int max_det) -> "
                   "(Tensor, Tensor, Tensor)")
    lib_def.define("batched_nms(Tensor boxes, Tensor scores, Tensor idxs, float iou_threshold) -> Tensor")
    impl = torch.library.Library("glsdet", "IMPL", "CUDA")
    cpu = torch.library.Library("glsdet", "IMPL", "CPU")
    meta = torch.library.Library("glsdet", "IMPL", "Meta")

    def no_cpu(*a, **k):
        raise RuntimeError("glsdet ops run on an MI355X only: there is no CPU fallback")

    # ------------------------------------------------------------------ conv_bn_act
    def conv_bn_act(x, w, scale, bias, cout, R, S, stride, pad, act, res=None, out_fp32=False):
        lib = _lib.load()
        _need_cuda(x, w, scale, bias, res)
        n, h, wd, c = x.shape
        ho, wo = (h + 2 * pad - R) // stride + 1, (wd + 2 * pad - S) // stride + 1
        ydt = torch.float32 if (out_fp32 or x.dtype == torch.float32) else x.dtype
        y = torch.empty(n, ho, wo, ceil_to(cout, 8), dtype=ydt, device=x.device)
        if w.dtype != x.dtype or not w.is_contiguous() or w.dim() != 2 or \
                w.shape[0] != lib.glsdet_conv_cout_pad(y.shape[3]) or w.shape[1] != lib.glsdet_conv_kpad(R, S, c, _DT[x.dtype]):
            raise RuntimeError("conv_bn_act: w must be pack_conv_weight(...)'s [cout_pad, kpad] matrix in x's dtype")
        if scale.dtype != torch.float32 or bias.dtype != torch.float32 or scale.numel() < w.shape[0] or bias.numel() < w.shape[0]:
            raise RuntimeError("conv_bn_act: scale / bias must be fp32 [cout_pad]")
        d = ConvDesc()
        d.x, d.y = _nhwc_view(x, "conv_bn_act.x"), _nhwc_view(y, "conv_bn_act.y")
        d.res = _nhwc_view(res, "conv_bn_act.res") if res is not None else View()
        d.w, d.scale, d.bias = w.data_ptr(), scale.data_ptr(), bias.data_ptr()
        d.R, d.S, d.stride, d.pad, d.act, d.tile_hint = R, S, stride, pad, act, 0
        check(lib.glsdet_conv2d(C.byref(d), _stream(x)), "conv_bn_act")
        return y

    def conv_bn_act_meta(x, w, scale, bias, cout, R, S, stride, pad, act, res=None, out_fp32=False):
        n, h, wd, c = x.shape
        ho, wo = (h + 2 * pad - R) // stride + 1, (wd + 2 * pad - S) // stride + 1
        return x.new_empty((n, ho, wo, ceil_to(cout, 8)), dtype=torch.float32 if out_fp32 else x.dtype)

    # ------------------------------------------------------------------ nonlocal_dot
    def nonlocal_dot(x, tpg, ci, wout, bout):
        lib = _lib.load()
        _need_cuda(x, tpg, wout, bout)
        if wout.dtype != torch.float32 or bout.dtype != torch.float32 or not wout.is_contiguous() or \
                tuple(wout.shape) != (x.shape[3], ci) or bout.numel() != x.shape[3] or tpg.shape[3] < 3 * ci:
            raise RuntimeError("nonlocal_dot: wout fp32 [C, ci], bout fp32 [C], tpg NHWC with >= 3*ci channels")
        y = torch.empty_like(x)
        ws = torch.empty(x.shape[0] * (8 * ci * ci + x.shape[3] * ci), dtype=torch.float32, device=x.device)
        check(lib.glsdet_nonlocal(C.byref(_nhwc_view(x, "nonlocal_dot.x")), C.byref(_nhwc_view(tpg, "nonlocal_dot.tpg")), ci,
                                  wout.data_ptr(), bout.data_ptr(), ws.data_ptr(), C.byref(_nhwc_view(y, "nonlocal_dot.y")),
                                  _stream(x)), "nonlocal_dot")
        return y

    # ------------------------------------------------------------------ yolox_decode
    def yolox_decode(levels, num_classes, in_h, in_w, mode=0):
        lib = _lib.load()
        _need_cuda(*levels)
        if any(l.dtype != torch.float32 for l in levels):
            raise RuntimeError("yolox_decode: the head logits are fp32 NHWC levels")
        n = levels[0].shape[0]
        A = sum(l.shape[1] * l.shape[2] for l in levels)
        out = torch.empty(n, A, 5 + num_classes, dtype=torch.float32, device=levels[0].device)
        arr = (View * len(levels))(*[_nhwc_view(l, "yolox_decode.level") for l in levels])
        strides = (C.c_int32 * len(levels))(*[in_h // l.shape[1] for l in levels]) if mode == 1 else None
        check(lib.glsdet_yolox_decode(arr, len(levels), num_classes, in_h, in_w, strides, mode, out.data_ptr(), out.numel(),
                                      None, _stream(out)), "yolox_decode")
        return out

    def yolox_decode_meta(levels, num_classes, in_h, in_w, mode=0):
        return levels[0].new_empty((levels[0].shape[0], sum(l.shape[1] * l.shape[2] for l in levels), 5 + num_classes))

    # ------------------------------------------------------------------ nms
    def nms(pred, num_classes, box_mode, conf_thres, nms_thres, max_det):
        lib = _lib.load()
        _need_cuda(pred)
        if pred.dtype != torch.float32 or not pred.is_contiguous() or pred.dim() != 3 or pred.shape[2] != 5 + num_classes:
            raise RuntimeError("nms: pred must be a contiguous fp32 [N, A, 5 + num_classes] tensor")
        n, A = pred.shape[0], pred.shape[1]
        ws = torch.empty(int(lib.glsdet_nms_workspace_bytes(n, A, A)), dtype=torch.uint8, device=pred.device)
        dets = torch.zeros(n, max_det, 7, dtype=torch.float32, device=pred.device)
        count = torch.zeros(2 * n, dtype=torch.int32, device=pred.device)
        status = torch.zeros(1, dtype=torch.int32, device=pred.device)
        check(lib.glsdet_nms(pred.data_ptr(), n, A, num_classes, box_mode, conf_thres, nms_thres, A, max_det, dets.data_ptr(),
                             count.data_ptr(), status.data_ptr(), ws.data_ptr(), ws.numel(), _stream(pred)), "nms")
        return dets, count, status

    def nms_meta(pred, num_classes, box_mode, conf_thres, nms_thres, max_det):
        n = pred.shape[0]
        return (pred.new_empty((n, max_det, 7)), pred.new_empty((2 * n,), dtype=torch.int32),
                pred.new_empty((1,), dtype=torch.int32))

    # ------------------------------------------------------------------ batched_nms (torchvision's signature)
    def batched_nms(boxes, scores, idxs, iou_threshold):
        _need_cuda(boxes, scores, idxs)
        m = boxes.shape[0]
        if m == 0:
            return torch.zeros(0, dtype=torch.int64, device=boxes.device)
        if (scores < 0).any():
            raise RuntimeError("batched_nms: scores must be >= 0")
        lab = idxs.to(torch.int64)
        nc = int(lab.max().item()) + 1
        pred = torch.full((1, m, 5 + nc), -1.0, dtype=torch.float32, device=boxes.device)
        pred[0, :, :4] = boxes.float()
        pred[0, :, 4] = 1.0
        pred[0, torch.arange(m, device=boxes.device), 5 + lab] = scores.float()
        dets, count, _ = nms(pred, nc, 1, -0.5, float(iou_threshold), m)
        k = int(count[0].item())
        kept = dets[0, :k]
        rows = torch.cat([boxes.float(), scores.float()[:, None], lab.float()[:, None]], 1)            # [m, 6]
        got = torch.cat([kept[:, :4], (kept[:, 4] * kept[:, 5])[:, None], kept[:, 6:7]], 1)             # [k, 6]
        same = (got[:, None, :] == rows[None, :, :]).all(-1)                                            # [k, m]
        return torch.argmax(same.to(torch.int8), dim=1)          # first (lowest) index of the identical row

    for name, fn, fm in (("conv_bn_act", conv_bn_act, conv_bn_act_meta), ("nonlocal_dot", nonlocal_dot, lambda x, *a: torch.empty_like(x)),
                         ("yolox_decode", yolox_decode, yolox_decode_meta), ("nms", nms, nms_meta),
                         ("batched_nms", batched_nms, lambda b, s, i, t: b.new_empty((0,), dtype=torch.int64))):
        impl.impl(name, fn)
        cpu.impl(name, no_cpu)
        meta.impl(name, fm)
    register._libs = (lib_def, impl, cpu, meta)          # keep the registrations alive


register()
