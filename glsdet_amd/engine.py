"""Host side of the HIP path: device memory, NHWC views, weight packing and the plan.

PyTorch is plumbing here (device allocations, streams); every arithmetic step of the
forward pass is a kernel of libglsdet_hip.so reached through the C ABI (``_lib``).
"""
from __future__ import annotations

import ctypes as C
import json
import os
import threading
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import F16, F32, ACT, View, ConvChain, ConvDesc, check

_TORCH_DT = {F16: torch.float16, F32: torch.float32}
_ESIZE = {F16: 2, F32: 4}


def ceil_to(v: int, m: int) -> int:
    return (v + m - 1) // m * m


class TView:
    """NHWC view into a device allocation (mirrors glsdet_view).  Strides in elements."""
    __slots__ = ("buf", "off", "n", "h", "w", "c", "sn", "sh", "sw", "dtype")

    def __init__(self, buf: torch.Tensor, off: int, n: int, h: int, w: int, c: int,
                 sn: int, sh: int, sw: int, dtype: int):
        self.buf, self.off = buf, off
        self.n, self.h, self.w, self.c = n, h, w, c
        self.sn, self.sh, self.sw, self.dtype = sn, sh, sw, dtype

    # ---- sub-views (no copies: this is how torch.cat / slicing of the reference is realised)
    def channels(self, c0: int, c1: int) -> "TView":
        assert 0 <= c0 < c1 <= self.c and c0 % 8 == 0, (c0, c1, self.c)
        return TView(self.buf, self.off + c0, self.n, self.h, self.w, c1 - c0, self.sn, self.sh, self.sw, self.dtype)

    def image(self, b: int) -> "TView":
        """the n = 1 view of image b"""
        assert 0 <= b < self.n
        return TView(self.buf, self.off + b * self.sn, 1, self.h, self.w, self.c, self.sn, self.sh, self.sw, self.dtype)

    def window(self, h0: int, h1: int, w0: int, w1: int) -> "TView":
        assert 0 <= h0 < h1 <= self.h and 0 <= w0 < w1 <= self.w
        return TView(self.buf, self.off + h0 * self.sh + w0 * self.sw, self.n, h1 - h0, w1 - w0, self.c,
                     self.sn, self.sh, self.sw, self.dtype)

    def as_c(self) -> View:
        es = _ESIZE[self.dtype]
        lo = self.buf.data_ptr()
        return View(lo + self.off * es, self.sn, self.sh, self.sw, self.n, self.h, self.w, self.c,
                    self.dtype, 0, lo, lo + self.buf.numel() * self.buf.element_size())

    def to_nchw(self, c: Optional[int] = None) -> torch.Tensor:
        """Materialise as a dense NCHW fp32 torch tensor (tests / the drop-in surface)."""
        t = torch.as_strided(self.buf.view(_TORCH_DT[self.dtype]), (self.n, self.h, self.w, self.c),
                             (self.sn, self.sh, self.sw, 1), self.off)
        t = t[..., : (c or self.c)]
        return t.permute(0, 3, 1, 2).float().contiguous()

    def __repr__(self):
        return "TView[%d,%d,%d,%d %s]" % (self.n, self.h, self.w, self.c, "f16" if self.dtype == F16 else "f32")


class Plan:
    """Recorded op sequence (glsdet_plan): eager replay, hipGraph capture/replay, timing."""

    def __init__(self, lib):
        self.lib = lib
        self.h = lib.glsdet_plan_create()
        self.captured = False

    def __del__(self):
        try:
            if self.h:
                self.lib.glsdet_plan_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def __enter__(self):
        check(self.lib.glsdet_plan_begin(self.h), "plan_begin")
        return self

    def __exit__(self, *exc):
        check(self.lib.glsdet_plan_end(self.h), "plan_end")
        return False

    @property
    def num_ops(self) -> int:
        return self.lib.glsdet_plan_num_ops(self.h)

    def ops(self):
        out = []
        name = C.create_string_buffer(160)
        kind, fl, by = C.c_int32(), C.c_double(), C.c_double()
        for i in range(self.num_ops):
            check(self.lib.glsdet_plan_op_info(self.h, i, C.byref(kind), C.byref(fl), C.byref(by), name, 160))
            out.append({"kind": kind.value, "flops": fl.value, "bytes": by.value, "name": name.value.decode()})
        return out

    def run(self, stream=None):
        check(self.lib.glsdet_plan_run(self.h, _stream_ptr(stream)), "plan_run")

    def capture(self, stream):
        check(self.lib.glsdet_plan_capture(self.h, _stream_ptr(stream)), "plan_capture")
        self.captured = True

    def launch(self, stream=None):
        check(self.lib.glsdet_plan_launch(self.h, _stream_ptr(stream)), "plan_launch")

    def run_timed(self, stream=None, reps: int = 1) -> np.ndarray:
        ms = (C.c_float * self.num_ops)()
        for _ in range(reps):
            check(self.lib.glsdet_plan_run_timed(self.h, _stream_ptr(stream), ms), "plan_run_timed")
        return np.asarray(ms[:], np.float64) / reps


def _stream_ptr(stream) -> int:
    if stream is None:
        stream = torch.cuda.current_stream()
    return stream.cuda_stream


_SHARED_TUNED: dict = {}
_TUNE_IO_LOCK = threading.Lock()       # save_tune_cache: lanes of one process share the table and the cache file


class _Ptr:
    """device address with the .data_ptr() of a tensor (a weight operand that lives in an activation buffer)"""

    def __init__(self, ptr: int):
        self._p = ptr

    def data_ptr(self) -> int:
        return self._p


class Engine:
    """Owns device buffers and packed weights; emits ops (eagerly or into a Plan)."""

    def __init__(self, dtype: str = "f16", device: str = "cuda:0", autotune: bool = False):
        self.lib = _lib.load()          # raises GlsdetLibraryError when the .so is missing
        if not torch.cuda.is_available():
            raise _lib.GlsdetLibraryError("glsdet_amd needs an MI355X visible to PyTorch-ROCm (no CPU fallback)")
        assert dtype in ("f16", "f32")
        self.dt = F16 if dtype == "f16" else F32
        self.device = torch.device(device)
        self.stream = None              # None -> torch current stream at call time
        self._keep: List[torch.Tensor] = []
        self.alloc_bytes = 0
        self.autotune = autotune        # measure kernel/tile variants per conv problem at build time
        # tuning results are shared by all engines of a dtype in this process: the second and third
        # plan instance of a detector reuse the first one's measurements (and pick identical kernels)
        self._tuned = _SHARED_TUNED.setdefault(dtype, {})
        # optional persistent tuning cache (JSON): GLSDET_TUNE_CACHE=/path/file.json
        self._tune_cache_path = os.environ.get("GLSDET_TUNE_CACHE", "")
        if self._tune_cache_path and os.path.exists(self._tune_cache_path):
            with open(self._tune_cache_path) as f:
                self._tuned.update({tuple(json.loads(k)): v for k, v in json.load(f).get(dtype, {}).items()})
        self._dtype_name = dtype
        self._vecs: dict = {}
        # verification (tools/variant_check.py): {"hint": h, "multi": h or None, "log": []} -- every conv also runs
        # on kernel variant h into a scratch tensor (same operands, before the real launch) and the two results are
        # compared element by element; eager emission only
        self.shadow = None

    # ---- memory
    def raw(self, nbytes: int) -> torch.Tensor:
        t = torch.zeros(ceil_to(max(nbytes, 16), 256), dtype=torch.uint8, device=self.device)
        self._keep.append(t)
        self.alloc_bytes += t.numel()
        return t

    def tensor(self, n: int, h: int, w: int, c: int, dtype: Optional[int] = None) -> TView:
        dt = self.dt if dtype is None else dtype
        cp = ceil_to(c, 8)
        buf = self.raw(n * h * w * cp * _ESIZE[dt])
        return TView(buf, 0, n, h, w, cp, h * w * cp, w * cp, cp, dt)

    def upload(self, arr: torch.Tensor) -> torch.Tensor:
        t = arr.contiguous().to(self.device)
        self._keep.append(t)
        self.alloc_bytes += t.numel() * t.element_size()
        return t

    # ---- weights
    def pack_conv(self, parts: Sequence[Tuple[torch.Tensor, torch.Tensor, torch.Tensor]], cin_pad: int):
        """parts: [(w OIHW fp32, scale[O], bias[O])...] concatenated along O.
        -> (w_dev [cout_pad][kpad] in engine dtype, scale_dev, bias_dev, cout, R, S)"""
        w = torch.cat([p[0].float() for p in parts], 0)
        scale = torch.cat([p[1].float() for p in parts], 0)
        bias = torch.cat([p[2].float() for p in parts], 0)
        cout, cin, R, S = w.shape
        assert cin <= cin_pad and cin_pad % 8 == 0
        kpad = self.lib.glsdet_conv_kpad(R, S, cin_pad, self.dt)
        cpad = self.lib.glsdet_conv_cout_pad(ceil_to(cout, 8))
        wp = torch.zeros(cpad, R, S, cin_pad, dtype=torch.float32)
        wp[:cout, :, :, :cin] = w.permute(0, 2, 3, 1)
        flat = torch.zeros(cpad, kpad, dtype=torch.float32)
        flat[:, : R * S * cin_pad] = wp.reshape(cpad, -1)
        sc = torch.ones(cpad, dtype=torch.float32)
        bi = torch.zeros(cpad, dtype=torch.float32)
        sc[:cout], bi[:cout] = scale, bias
        return (self.upload(flat.to(_TORCH_DT[self.dt])), self.upload(sc), self.upload(bi), cout, R, S)

    # ---- ops (each is one C-ABI call)
    def conv(self, x: TView, packed, stride: int, pad: int, act: str, out: Optional[TView] = None,
             res: Optional[TView] = None, out_dtype: Optional[int] = None, tile_hint: int = 0,
             res_first: bool = False) -> TView:
        """res_first: act(conv*scale + bias + res) (ResNet) instead of act(conv*scale + bias) + res."""
        wdev, sdev, bdev, cout, R, S = packed
        ho = (x.h + 2 * pad - R) // stride + 1
        wo = (x.w + 2 * pad - S) // stride + 1
        if out is None:
            out = self.tensor(x.n, ho, wo, cout, out_dtype)
        assert out.c == ceil_to(cout, 8), (out.c, cout)
        d = ConvDesc()
        d.x, d.y = x.as_c(), out.as_c()
        d.res = res.as_c() if res is not None else View()
        d.w, d.scale, d.bias = wdev.data_ptr(), sdev.data_ptr(), bdev.data_ptr()
        d.R, d.S, d.stride, d.pad, d.act, d.tile_hint = R, S, stride, pad, ACT[act], tile_hint
        if res_first and res is not None:
            d.act |= 0x100                      # GLSDET_ACT_RES_FIRST
        if self.autotune and tile_hint == 0:
            key = (x.n, x.h, x.w, x.c, x.sn, x.sh, x.sw, out.c, out.sn, out.sh, out.sw, R, S, stride, pad,
                   res is not None, out.dtype)
            if key not in self._tuned:
                best, us = C.c_int32(0), C.c_float(0)
                check(self.lib.glsdet_conv2d_tune(C.byref(d), _stream_ptr(self.stream), C.byref(best), C.byref(us)),
                      "conv2d_tune")
                self._tuned[key] = best.value
                self._tune_dirty = True
            d.tile_hint = self._tuned[key]
        sh = None
        if self.shadow is not None and tile_hint == 0 and self.shadow.get("hint"):
            sh = self._shadow_begin([d], [out], self.shadow["hint"], multi=False)
        forced = os.environ.get("GLSDET_FORCE_HINT")          # verification runs: every conv uses this variant
        if forced and tile_hint == 0:                          # wherever it accepts the problem
            d.tile_hint = int(forced, 0)
            if self.lib.glsdet_conv2d(C.byref(d), _stream_ptr(self.stream)) == 0:
                return out
            d.tile_hint = 0
        check(self.lib.glsdet_conv2d(C.byref(d), _stream_ptr(self.stream)), "conv2d")
        self._shadow_end(sh)
        return out

    def conv_chain(self, x: TView, packed, stride: int, pad: int, act: str, out: TView, res: Optional[TView],
                   packed2, act2: str, c0: int, cin2: int, out2: TView, res_first: bool = False, skip_y: bool = False) -> bool:
        """conv (as Engine.conv) + in the same launch a 1x1 conv (packed2, act2) on channels [c0, c0 + cin2) of its result
        -> out2 (glsdet_conv2d_chain).  Returns False, having launched nothing, when no kernel takes the fused problem
        (the caller then emits the two convs).  skip_y: nothing else reads `out` -- it is not stored (GLSDET_CHAIN_SKIP_Y;
        `out` is then only a scratch view for the tuner's comparison and the shadow check)."""
        wdev, sdev, bdev, cout, R, S = packed
        w2, s2, b2, cout2, R2, S2 = packed2
        assert R2 == 1 and S2 == 1 and out2.c == ceil_to(cout2, 8)
        d = ConvDesc()
        d.x, d.y = x.as_c(), out.as_c()
        d.res = res.as_c() if res is not None else View()
        d.w, d.scale, d.bias = wdev.data_ptr(), sdev.data_ptr(), bdev.data_ptr()
        d.R, d.S, d.stride, d.pad, d.act, d.tile_hint = R, S, stride, pad, ACT[act], 0
        if res_first and res is not None:
            d.act |= 0x100
        c = ConvChain()
        c.y2 = out2.as_c()
        c.w2, c.scale2, c.bias2 = w2.data_ptr(), s2.data_ptr(), b2.data_ptr()
        c.act2, c.c0, c.cin2 = ACT[act2], c0, cin2
        c.flags = 1 if skip_y else 0
        st = _stream_ptr(self.stream)
        if self.autotune:
            key = ("chain", x.n, x.h, x.w, x.c, x.sn, x.sh, x.sw, out.c, out.sn, out.sh, out.sw, R, S, stride, pad,
                   res is not None, out.dtype, c0, c.cin2, out2.c, out2.sn, out2.sh, out2.sw, bool(skip_y))
            if key not in self._tuned:
                best, us = C.c_int32(0), C.c_float(0)
                rc = self.lib.glsdet_conv2d_chain_tune(C.byref(d), C.byref(c), st, C.byref(best), C.byref(us))
                hint = best.value if rc == 0 else -1
                if hint >= 0:
                    # fused only where it beats the two launches it replaces (each tuned on its own): the chained product
                    # runs in the tail of the producing workgroups, which costs more than a launch on some shapes
                    b1, u1, b2h, u2 = C.c_int32(0), C.c_float(0), C.c_int32(0), C.c_float(0)
                    d2 = ConvDesc()
                    d2.x, d2.y, d2.res = out.channels(c0, c0 + cin2).as_c(), out2.as_c(), View()
                    d2.w, d2.scale, d2.bias = w2.data_ptr(), s2.data_ptr(), b2.data_ptr()
                    d2.R, d2.S, d2.stride, d2.pad, d2.act, d2.tile_hint = 1, 1, 1, 0, ACT[act2], 0
                    if self.lib.glsdet_conv2d_tune(C.byref(d), st, C.byref(b1), C.byref(u1)) == 0 and \
                            self.lib.glsdet_conv2d_tune(C.byref(d2), st, C.byref(b2h), C.byref(u2)) == 0 and \
                            us.value > 0.97 * (u1.value + u2.value) + _fuse_credit_us():
                        hint = -1
                self._tuned[key] = hint
                self._tune_dirty = True
            if self._tuned[key] < 0:
                return False
            d.tile_hint = self._tuned[key]
        sh = None
        if self.shadow is not None and self.shadow.get("hint") and not skip_y:   # the variant under test computes y (unfused) first
            sh = self._shadow_begin([d], [out], self.shadow["hint"], multi=False)
        ok = self.lib.glsdet_conv2d_chain(C.byref(d), C.byref(c), st) == 0
        if ok and self.shadow is not None:
            self._shadow_end(sh)
            if skip_y:                                # y was not stored: produce it for the stand-alone 1x1 of the check below
                self.conv(x, packed, stride, pad, act, out=out, tile_hint=1)
            # and the chained product against a stand-alone 1x1 on the stored y: bit for bit
            ref = self.tensor(out2.n, out2.h, out2.w, out2.c, out2.dtype)
            self.conv(out.channels(c0, c0 + cin2), packed2, 1, 0, act2, out=ref, tile_hint=1)
            a, b = out2.to_nchw(), ref.to_nchw()
            self.shadow["log"].append({"shape": (x.n, out.h, out.w, cin2, out2.c, 1, 1), "scale": float(a.abs().max()),
                                       "err": float((a - b).abs().max()), "nan": bool(torch.isnan(a).any()),
                                       "differ": float((a != b).float().mean())})
        return ok

    def bottleneck(self, x: TView, packed1, act1: str, packed2, act2: str, out: TView, res: Optional[TView],
                   hidden: TView) -> bool:
        """y = act2(conv3x3(act1(conv1x1(x)))) (+ res) in ONE launch (glsdet_bottleneck): the 1x1 is recomputed on the
        halo of the 3x3's tiles and the hidden tensor never exists in memory.  `out` must not alias x.  Returns False,
        having launched nothing, when the fused kernel does not apply or (autotune) does not beat the two tuned launches
        it replaces; `hidden` is the scratch tensor the comparison (and the caller's fallback) uses."""
        w1, s1, b1, cm, R1, S1 = packed1
        w2, s2, b2, cout, R2, S2 = packed2
        if (R1, S1, R2, S2) != (1, 1, 3, 3) or cm != cout or cm not in (32, 64, 128) or x.dtype != out.dtype:
            return False
        hid = hidden
        d1, d2 = ConvDesc(), ConvDesc()
        d1.x, d1.y, d1.res = x.as_c(), hid.as_c(), View()
        d1.w, d1.scale, d1.bias = w1.data_ptr(), s1.data_ptr(), b1.data_ptr()
        d1.R, d1.S, d1.stride, d1.pad, d1.act, d1.tile_hint = 1, 1, 1, 0, ACT[act1], 0
        d2.x, d2.y = hid.as_c(), out.as_c()
        d2.res = res.as_c() if res is not None else View()
        d2.w, d2.scale, d2.bias = w2.data_ptr(), s2.data_ptr(), b2.data_ptr()
        d2.R, d2.S, d2.stride, d2.pad, d2.act, d2.tile_hint = 3, 3, 1, 1, ACT[act2], 0
        st = _stream_ptr(self.stream)
        hint = 0
        if self.autotune:
            key = ("bneck", x.n, x.h, x.w, x.c, x.sn, x.sh, x.sw, cm, out.sn, out.sh, out.sw, res is not None, out.dtype)
            if key not in self._tuned:
                best, us = C.c_int32(0), C.c_float(0)
                hint = best.value if self.lib.glsdet_bottleneck_tune(C.byref(d1), C.byref(d2), st, C.byref(best), C.byref(us)) == 0 else -1
                if hint >= 0:
                    hint = best.value
                    # against the two launches it replaces, each tuned on its own
                    h1, u1, h2, u2 = C.c_int32(0), C.c_float(0), C.c_int32(0), C.c_float(0)
                    if self.lib.glsdet_conv2d_tune(C.byref(d1), st, C.byref(h1), C.byref(u1)) == 0 and \
                            self.lib.glsdet_conv2d_tune(C.byref(d2), st, C.byref(h2), C.byref(u2)) == 0 and \
                            us.value > 0.97 * (u1.value + u2.value) + _fuse_credit_us() and not os.environ.get("GLSDET_FORCE_BNECK"):
                        hint = -1
                self._tuned[key] = hint
                self._tune_dirty = True
            hint = self._tuned[key]
            if hint < 0:
                return False
        ok = self.lib.glsdet_bottleneck(C.byref(d1), C.byref(d2), hint, st) == 0
        if ok and self.shadow is not None:      # against the two-launch form on the same operands: bit for bit
            ref = self.tensor(out.n, out.h, out.w, out.c, out.dtype)
            self.conv(x, packed1, 1, 0, act1, out=hidden, tile_hint=1)
            self.conv(hidden, packed2, 1, 1, act2, out=ref, res=res, tile_hint=1)
            a, b = out.to_nchw(), ref.to_nchw()
            self.shadow["log"].append({"shape": (x.n, out.h, out.w, cm, cm, 3, 3), "scale": float(a.abs().max()),
                                       "err": float((a - b).abs().max()), "nan": bool(torch.isnan(a).any()),
                                       "differ": float((a != b).float().mean())})
        return ok

    def conv_multi(self, xs: Sequence[TView], packs, stride: int, pad: int, act: str,
                   outs: Optional[Sequence[Optional[TView]]] = None, ress: Optional[Sequence[Optional[TView]]] = None,
                   out_dtype: Optional[int] = None, tile_hint: int = 0) -> List[TView]:
        """Up to 8 independent convs of one shape class (same k, stride, Cin, Cout) as ONE launch
        (glsdet_conv2d_multi): each alone is too small to fill the chip."""
        n = len(xs)
        assert 1 <= n <= 32 and len(packs) == n
        outs = list(outs) if outs is not None else [None] * n
        ress = list(ress) if ress is not None else [None] * n
        arr = (ConvDesc * n)()
        for i, (x, pk) in enumerate(zip(xs, packs)):
            wdev, sdev, bdev, cout, R, S = pk
            if outs[i] is None:
                outs[i] = self.tensor(x.n, (x.h + 2 * pad - R) // stride + 1, (x.w + 2 * pad - S) // stride + 1, cout, out_dtype)
            d = arr[i]
            d.x, d.y = x.as_c(), outs[i].as_c()
            d.res = ress[i].as_c() if ress[i] is not None else View()
            d.w, d.scale, d.bias = wdev.data_ptr(), sdev.data_ptr(), bdev.data_ptr()
            d.R, d.S, d.stride, d.pad, d.act, d.tile_hint = R, S, stride, pad, ACT[act], tile_hint
        sh = None
        if self.shadow is not None and self.shadow.get("multi"):
            sh = self._shadow_begin([arr[i] for i in range(n)], outs, self.shadow["multi"], multi=True)
        forced = os.environ.get("GLSDET_FORCE_MULTI_HINT")
        if forced and tile_hint == 0:
            arr[0].tile_hint = int(forced, 0)
            if self.lib.glsdet_conv2d_multi(arr, n, _stream_ptr(self.stream)) == 0:
                return outs
            arr[0].tile_hint = 0
        check(self.lib.glsdet_conv2d_multi(arr, n, _stream_ptr(self.stream)), "conv2d_multi")
        self._shadow_end(sh)
        return outs

    def conv_many(self, xs: Sequence[TView], packs, stride: int, pad: int, act: str, outs: Sequence[TView],
                  ress: Optional[Sequence[Optional[TView]]] = None) -> List[TView]:
        """Any number of independent convs of one shape class: eight per launch, or -- 1x1 problems of ONE geometry (sizes,
        strides, residual or not: the per-window GEMMs of the ResNet GL plug-in) -- up to 32 per launch (the batched form of
        glsdet_conv2d_multi: one argument block + the operand addresses of each problem), on the tile the tuner measures."""
        ress = list(ress) if ress is not None else [None] * len(xs)
        n = len(xs)
        geo = lambda x, pk, o, r: (x.n, x.h, x.w, x.c, x.sn, x.sh, x.sw, x.dtype, pk[3], pk[4], pk[5], o.n, o.h, o.w, o.c, o.sn, o.sh,
                                   o.sw, o.dtype) + ((-1, -1, -1, -1) if r is None else (r.sn, r.sh, r.sw, r.c))   # (flat: a tune-cache key)
        per = 8
        if n > 8 and stride == 1 and pad == 0 and packs[0][4] == 1 and packs[0][5] == 1 and not os.environ.get("GLSDET_NO_BATCH") and \
                len({geo(x, pk, o, r) for x, pk, o, r in zip(xs, packs, outs, ress)}) == 1:
            per = 32
        for i in range(0, n, per):
            k = min(per, n - i)
            hint = 0
            if k > 8 and self.autotune:
                key = ("batch", k, act) + geo(xs[i], packs[i], outs[i], ress[i])
                if key not in self._tuned:
                    arr = (ConvDesc * k)()
                    for j in range(k):
                        d, pk, r = arr[j], packs[i + j], ress[i + j]
                        d.x, d.y = xs[i + j].as_c(), outs[i + j].as_c()
                        d.res = r.as_c() if r is not None else View()
                        d.w, d.scale, d.bias = pk[0].data_ptr(), pk[1].data_ptr(), pk[2].data_ptr()
                        d.R, d.S, d.stride, d.pad, d.act, d.tile_hint = 1, 1, 1, 0, ACT[act], 0
                    best, us = C.c_int32(0), C.c_float(0)
                    check(self.lib.glsdet_conv2d_multi_tune(arr, k, _stream_ptr(self.stream), C.byref(best), C.byref(us)),
                          "conv2d_multi_tune")
                    self._tuned[key] = best.value
                    self._tune_dirty = True
                hint = self._tuned[key]
            self.conv_multi(xs[i:i + k], packs[i:i + k], stride, pad, act, outs=outs[i:i + k], ress=ress[i:i + k], tile_hint=hint)
        return list(outs)

    def matrix(self, rows: int, cols: int, dtype: Optional[int] = None) -> TView:
        """Dense rows x cols matrix that can serve BOTH as a 1x1-conv activation (rows = pixels, cols = channels)
        and as a 1x1-conv weight operand (rows = output channels): row pitch = glsdet_conv_kpad(cols), rows padded
        to the weight tile granule (32), zero filled.  -> view [1, 1, rows, ceil8(cols)]."""
        dt = self.dt if dtype is None else dtype
        c8 = ceil_to(cols, 8)
        pitch = self.lib.glsdet_conv_kpad(1, 1, c8, dt)
        rp = self.lib.glsdet_conv_cout_pad(ceil_to(rows, 8))
        buf = self.raw(rp * pitch * _ESIZE[dt])
        return TView(buf, 0, 1, 1, rows, c8, rows * pitch, rows * pitch, pitch, dt)

    def bias_vector(self, n: int) -> TView:
        """fp32 vector of n entries (zero filled, padded to the weight row granule) that a 1-pixel conv of the plan writes
        and a later conv reads as its per-channel bias (as_weight(bias_dev=)): -> view [1, 1, 1, n]."""
        cpad = self.lib.glsdet_conv_cout_pad(ceil_to(n, 8))
        buf = self.raw(cpad * 4)
        return TView(buf, 0, 1, 1, 1, ceil_to(n, 8), cpad, cpad, cpad, F32)

    def as_weight(self, m: TView, alpha: float = 1.0, bias: Optional[torch.Tensor] = None, bias_dev: Optional[TView] = None):
        """`packed` tuple for Engine.conv whose weight operand is the activation matrix m (Engine.matrix layout):
        out[pixel][r] = alpha * sum_k x[pixel][k] * m[r][k] (+ bias[r]).  bias: host constant; bias_dev: an fp32 vector an
        earlier op of the plan writes (Engine.bias_vector)."""
        rows = m.h * m.w
        assert m.n == 1 and m.sw == self.lib.glsdet_conv_kpad(1, 1, m.c, m.dtype), "as_weight needs an Engine.matrix view"
        cpad = self.lib.glsdet_conv_cout_pad(ceil_to(rows, 8))
        if bias_dev is not None:
            assert bias is None and bias_dev.dtype == F32 and bias_dev.sw >= cpad, "bias_dev: Engine.bias_vector of >= cout_pad floats"
            key = ("vec", cpad, float(alpha))
            if key not in self._vecs:
                self._vecs[key] = (self.upload(torch.full((cpad,), float(alpha))), self.upload(torch.zeros(cpad)))
            return (_Ptr(m.buf.data_ptr() + m.off * _ESIZE[m.dtype]), self._vecs[key][0],
                    _Ptr(bias_dev.buf.data_ptr() + bias_dev.off * _ESIZE[F32]), ceil_to(rows, 8), 1, 1)
        key = ("vec", cpad, float(alpha))
        if key not in self._vecs:
            self._vecs[key] = (self.upload(torch.full((cpad,), float(alpha))), self.upload(torch.zeros(cpad)))
        sc, zero = self._vecs[key]
        if bias is not None:
            bkey = ("bias", cpad, bias.data_ptr())
            if bkey not in self._vecs:
                b = torch.zeros(cpad)
                b[: bias.numel()] = bias.float()
                self._vecs[bkey] = (self.upload(b), bias)        # keeps `bias` alive: its address is the key
            zero = self._vecs[bkey][0]
        return (_Ptr(m.buf.data_ptr() + m.off * _ESIZE[m.dtype]), sc, zero, ceil_to(rows, 8), 1, 1)

    def conv_group(self, xs: Sequence[TView], packs, stride: int, pad: int, act: str,
                   outs: Optional[Sequence[Optional[TView]]] = None, ress: Optional[Sequence[Optional[TView]]] = None,
                   out_dtype: Optional[int] = None) -> List[TView]:
        """n independent convs of one shape class: ONE grouped launch when that is faster than n
        launches (measured at build time with autotune, a size heuristic without), else n convs."""
        n = len(xs)
        outs = list(outs) if outs is not None else [None] * n
        ress = list(ress) if ress is not None else [None] * n
        R, S, cout = packs[0][4], packs[0][5], packs[0][3]
        for i, x in enumerate(xs):
            if outs[i] is None:
                outs[i] = self.tensor(x.n, (x.h + 2 * pad - R) // stride + 1, (x.w + 2 * pad - S) // stride + 1, cout, out_dtype)
        separate = lambda: [self.conv(x, pk, stride, pad, act, out=o, res=r, out_dtype=out_dtype)
                            for x, pk, o, r in zip(xs, packs, outs, ress)]
        if n < 2 or n > 4 or len({(pk[3], pk[4], pk[5], x.c) for x, pk in zip(xs, packs)}) != 1 or \
                os.environ.get("GLSDET_NO_GROUP"):          # A/B switch for measurements
            return separate()
        if not self.autotune:
            small = all(o.n * o.h * o.w * o.c <= 64 * 64 * 384 for o in outs)
            return self.conv_multi(xs, packs, stride, pad, act, outs=outs, ress=ress) if small else separate()
        key = ("multi", n, stride, pad, R, S) + tuple(v for x, o, r in zip(xs, outs, ress) for v in (
            x.n, x.h, x.w, x.c, x.sn, x.sh, x.sw, o.c, o.sn, o.sh, o.sw, r is not None, o.dtype))
        if key not in self._tuned:
            arr = (ConvDesc * n)()
            single_us = 0.0
            for i, (x, pk, o, r) in enumerate(zip(xs, packs, outs, ress)):
                d = arr[i]
                d.x, d.y = x.as_c(), o.as_c()
                d.res = r.as_c() if r is not None else View()
                d.w, d.scale, d.bias = pk[0].data_ptr(), pk[1].data_ptr(), pk[2].data_ptr()
                d.R, d.S, d.stride, d.pad, d.act, d.tile_hint = R, S, stride, pad, ACT[act], 0
                best, us = C.c_int32(0), C.c_float(0)
                check(self.lib.glsdet_conv2d_tune(C.byref(d), _stream_ptr(self.stream), C.byref(best), C.byref(us)), "conv2d_tune")
                single_us += us.value
            best, us = C.c_int32(0), C.c_float(0)
            check(self.lib.glsdet_conv2d_multi_tune(arr, n, _stream_ptr(self.stream), C.byref(best), C.byref(us)),
                  "conv2d_multi_tune")
            self._tuned[key] = best.value if us.value < 0.95 * single_us + (n - 1) * _fuse_credit_us() else -1
            self._tune_dirty = True
        hint = self._tuned[key]
        if hint < 0:
            return separate()
        return self.conv_multi(xs, packs, stride, pad, act, outs=outs, ress=ress, tile_hint=hint)

    def _shadow_begin(self, descs, outs, hint: int, multi: bool):
        """Run the conv(s) on kernel variant `hint` into dense scratch tensors BEFORE the real launch (an in-place
        conv still sees its original residual).  -> state for _shadow_end, or None if the variant declines."""
        n = len(descs)
        arr = (ConvDesc * n)()
        tmps = []
        for i, (d, o) in enumerate(zip(descs, outs)):
            C.memmove(C.addressof(arr[i]), C.addressof(d), C.sizeof(ConvDesc))
            t = self.tensor(o.n, o.h, o.w, o.c, o.dtype)
            tmps.append(t)
            arr[i].y = t.as_c()
            arr[i].tile_hint = hint
        st = _stream_ptr(self.stream)
        rc = self.lib.glsdet_conv2d_multi(arr, n, st) if multi else self.lib.glsdet_conv2d(C.byref(arr[0]), st)
        if rc != 0:                       # the variant does not take this problem
            return None
        return [(d.x.n, d.x.h, d.x.w, d.x.c, o.c, d.R, d.stride) for d, o in zip(descs, outs)], list(outs), tmps

    def _shadow_end(self, state):
        """After the real launch: element-wise distance of the variant's result from it."""
        if state is None:
            return
        for shape, o, t in zip(*state):
            a, b = o.to_nchw(), t.to_nchw()
            self.shadow["log"].append({"shape": shape, "scale": float(a.abs().max()), "err": float((a - b).abs().max()),
                                       "nan": bool(torch.isnan(b).any()), "differ": float((a != b).float().mean())})

    def pack_dw(self, w: torch.Tensor, scale: torch.Tensor, bias: torch.Tensor, c_pad: int):
        """depthwise weights [C,1,R,S] -> ([R*S][c_pad] in engine dtype, scale, bias, C, R, S)"""
        C_, one, R, S = w.shape
        assert one == 1 and C_ <= c_pad and c_pad % 8 == 0
        wp = torch.zeros(R * S, c_pad, dtype=torch.float32)
        wp[:, :C_] = w.float().reshape(C_, R * S).t()
        sc, bi = torch.ones(c_pad), torch.zeros(c_pad)
        sc[:C_], bi[:C_] = scale.float(), bias.float()
        return (self.upload(wp.to(_TORCH_DT[self.dt])), self.upload(sc), self.upload(bi), C_, R, S)

    def gate(self, a: TView, b: TView, gmap: Optional[TView], out: Optional[TView] = None) -> TView:
        """glsdet_gate: a * map[0] + b * map[1] with a map, a * b without"""
        if out is None:
            out = self.tensor(a.n, a.h, a.w, a.c, a.dtype)
        check(self.lib.glsdet_gate(C.byref(a.as_c()), C.byref(b.as_c()), C.byref(gmap.as_c()) if gmap is not None else None,
                                   C.byref(out.as_c()), 0 if gmap is not None else 1, _stream_ptr(self.stream)), "gate")
        return out

    def dwconv(self, x: TView, packed, stride: int, pad: int, act: str, out: Optional[TView] = None, dilation: int = 1) -> TView:
        wdev, sdev, bdev, C_, R, S = packed
        ho = (x.h + 2 * pad - dilation * (R - 1) - 1) // stride + 1
        wo = (x.w + 2 * pad - dilation * (S - 1) - 1) // stride + 1
        if out is None:
            out = self.tensor(x.n, ho, wo, x.c, x.dtype)
        d = ConvDesc()
        d.x, d.y, d.res = x.as_c(), out.as_c(), View()
        d.w, d.scale, d.bias = wdev.data_ptr(), sdev.data_ptr(), bdev.data_ptr()
        d.R, d.S, d.stride, d.pad, d.act, d.tile_hint = R, S, stride, pad, ACT[act], 0
        check(self.lib.glsdet_dwconv2d_dilated(C.byref(d), dilation, _stream_ptr(self.stream)), "dwconv2d")
        return out

    def focus_pack(self, img: torch.Tensor, out: Optional[TView] = None) -> TView:
        assert img.dtype == torch.float32 and img.is_contiguous() and img.device.type == "cuda"
        n, cin, H, W = img.shape
        if out is None:
            out = self.tensor(n, H // 2, W // 2, ceil_to(4 * cin, 8))
        check(self.lib.glsdet_focus_pack(img.data_ptr(), n, cin, H, W, C.byref(out.as_c()),
                                         _stream_ptr(self.stream)), "focus_pack")
        return out

    def focus_conv(self, img: torch.Tensor, packed, act: str, out: Optional[TView] = None) -> TView:
        """Focus + its 3x3 conv in one launch (glsdet_focus_conv); packed = pack_conv(..., cin_pad=16)."""
        assert img.dtype == torch.float32 and img.is_contiguous() and img.device.type == "cuda"
        n, cin, H, W = img.shape
        wdev, sdev, bdev, cout, R, S = packed
        assert (R, S) == (3, 3)
        if out is None:
            out = self.tensor(n, H // 2, W // 2, cout)
        check(self.lib.glsdet_focus_conv(img.data_ptr(), n, cin, H, W, wdev.data_ptr(), sdev.data_ptr(), bdev.data_ptr(),
                                         ACT[act], C.byref(out.as_c()), _stream_ptr(self.stream)), "focus_conv")
        return out

    def focus_conv_down(self, img: torch.Tensor, packed1, act1: str, packed2, act2: str, out: Optional[TView] = None) -> TView:
        """Focus + stem conv + the 3x3 stride-2 conv that follows, in one launch (glsdet_focus_conv_down)."""
        assert img.dtype == torch.float32 and img.is_contiguous() and img.device.type == "cuda"
        n, cin, H, W = img.shape
        w1, s1, b1, c1, R1, S1 = packed1
        w2, s2, b2, cout, R2, S2 = packed2
        assert (R1, S1, R2, S2) == (3, 3, 3, 3)
        if out is None:
            out = self.tensor(n, (H // 2 + 1) // 2, (W // 2 + 1) // 2, cout)
        check(self.lib.glsdet_focus_conv_down(img.data_ptr(), n, cin, H, W, w1.data_ptr(), s1.data_ptr(), b1.data_ptr(), ACT[act1], c1,
                                              w2.data_ptr(), s2.data_ptr(), b2.data_ptr(), ACT[act2], C.byref(out.as_c()),
                                              _stream_ptr(self.stream)), "focus_conv_down")
        return out

    def channel_maxmean(self, x: TView, out: Optional[TView] = None) -> TView:
        if out is None:
            out = self.tensor(x.n, x.h, x.w, 8, x.dtype)
        check(self.lib.glsdet_channel_maxmean(C.byref(x.as_c()), C.byref(out.as_c()), _stream_ptr(self.stream)),
              "channel_maxmean")
        return out

    def maxpool(self, x: TView, k: int, out: Optional[TView] = None) -> TView:
        if out is None:
            out = self.tensor(x.n, x.h, x.w, x.c, x.dtype)
        check(self.lib.glsdet_maxpool2d(C.byref(x.as_c()), C.byref(out.as_c()), k, _stream_ptr(self.stream)), "maxpool2d")
        return out

    def spp_pools(self, x: TView, y5: TView, y9: TView, y13: TView):
        check(self.lib.glsdet_spp_pools(C.byref(x.as_c()), C.byref(y5.as_c()), C.byref(y9.as_c()), C.byref(y13.as_c()),
                                        _stream_ptr(self.stream)), "spp_pools")

    def resample(self, x: TView, factor: int, out: Optional[TView] = None) -> TView:
        if out is None:
            out = self.tensor(x.n, x.h * factor, x.w * factor, x.c, x.dtype)
        check(self.lib.glsdet_resample_copy(C.byref(x.as_c()), C.byref(out.as_c()), factor,
                                            _stream_ptr(self.stream)), "resample_copy")
        return out

    def copy_many(self, xs: Sequence[TView], outs: Sequence[TView]) -> List[TView]:
        """xs[i] -> outs[i] (strided copies of equal extents, one dtype / channel count) in one launch per 32 pairs."""
        n = len(xs)
        assert n == len(outs) and n > 0
        xa = (View * n)(*[x.as_c() for x in xs])
        ya = (View * n)(*[y.as_c() for y in outs])
        check(self.lib.glsdet_copy_many(xa, ya, n, _stream_ptr(self.stream)), "copy_many")
        return list(outs)

    def transpose_many(self, xs: Sequence[TView], mats: Sequence[TView]) -> List[TView]:
        """mats[i][c][p] = xs[i] at pixel p, channel c (xs[i]: one image window, mats[i]: Engine.matrix with >= C rows and
        >= h*w columns): one launch per 32 pairs."""
        n = len(xs)
        assert n == len(mats) and n > 0
        xa = (View * n)(*[x.as_c() for x in xs])
        ya = (View * n)(*[y.as_c() for y in mats])
        check(self.lib.glsdet_transpose_many(xa, ya, n, _stream_ptr(self.stream)), "transpose_many")
        return list(mats)

    def nonlocal_(self, x: TView, tpg: TView, ci: int, wout: torch.Tensor, bout: torch.Tensor,
                  out: Optional[TView] = None) -> TView:
        if out is None:
            out = self.tensor(x.n, x.h, x.w, x.c, x.dtype)
        ws = self.raw(x.n * (8 * ci * ci + x.c * ci) * 4)   # 8 partial Gram slices + folded P
        check(self.lib.glsdet_nonlocal(C.byref(x.as_c()), C.byref(tpg.as_c()), ci, wout.data_ptr(), bout.data_ptr(),
                                       ws.data_ptr(), C.byref(out.as_c()), _stream_ptr(self.stream)), "nonlocal")
        return out

    def nonlocal_multi(self, xs: Sequence[TView], tpgs: Sequence[TView], ci: int, wouts, bouts) -> List[TView]:
        """1..4 independent non-local blocks (in place on each x) in one set of three launches."""
        n = len(xs)
        ws = self.raw(n * xs[0].n * (8 * ci * ci + xs[0].c * ci) * 4)
        xa = (View * n)(*[x.as_c() for x in xs])
        ta = (View * n)(*[t.as_c() for t in tpgs])
        wa = (C.c_void_p * n)(*[w.data_ptr() for w in wouts])
        ba = (C.c_void_p * n)(*[b.data_ptr() for b in bouts])
        check(self.lib.glsdet_nonlocal_multi(xa, ta, n, ci, wa, ba, ws.data_ptr(), xa, _stream_ptr(self.stream)),
              "nonlocal_multi")
        return list(xs)

    # ---- Patch_Conv_NonLocal_adapt_new: device-side quadrant split
    def attn_split(self, att: TView) -> torch.Tensor:
        """-> device int32[4] {row split, column split above it, column split from it on, 0} (glsdet_attn_split)."""
        split = torch.zeros(4, dtype=torch.int32, device=self.device)
        self._keep.append(split)
        check(self.lib.glsdet_attn_split(C.byref(att.as_c()), split.data_ptr(), _stream_ptr(self.stream)), "attn_split")
        return split

    def nonlocal_split(self, x: TView, tpgs: Sequence[TView], ci: int, wouts, bouts, out: TView, split: torch.Tensor,
                       shift: int = 0) -> TView:
        ws = self.raw(4 * x.n * (8 * ci * ci + x.c * ci) * 4)
        ta = (View * 4)(*[t.as_c() for t in tpgs])
        wa = (C.c_void_p * 4)(*[w.data_ptr() for w in wouts])
        ba = (C.c_void_p * 4)(*[b.data_ptr() for b in bouts])
        check(self.lib.glsdet_nonlocal_split(C.byref(x.as_c()), ta, ci, wa, ba, ws.data_ptr(), C.byref(out.as_c()),
                                             split.data_ptr(), shift, _stream_ptr(self.stream)), "nonlocal_split")
        return out

    def rowsplit(self, a: TView, b: Optional[TView], split: torch.Tensor, mode: int, out: Optional[TView] = None, quadrant: int = 0,
                 shift: int = 0) -> TView:
        """glsdet_rowsplit: 0 / 1 keep the rows above / from the split, 2 select a | b by row, 3 keep quadrant, 4 merge quadrant"""
        if out is None:
            out = self.tensor(a.n, a.h, a.w, a.c, a.dtype)
        check(self.lib.glsdet_rowsplit(C.byref(a.as_c()), C.byref(b.as_c()) if b is not None else None, C.byref(out.as_c()),
                                       split.data_ptr(), mode | (quadrant << 4) | (shift << 8), _stream_ptr(self.stream)), "rowsplit")
        return out

    def scale_by_map(self, x: TView, m: TView, out: Optional[TView] = None) -> TView:
        if out is None:
            out = self.tensor(x.n, x.h, x.w, x.c, x.dtype)
        check(self.lib.glsdet_scale_by_map(C.byref(x.as_c()), C.byref(m.as_c()), C.byref(out.as_c()), _stream_ptr(self.stream)),
              "scale_by_map")
        return out

    def decode(self, levels: Sequence[TView], num_classes: int, in_h: int, in_w: int,
               strides: Optional[Sequence[int]] = None, mode: int = 0, out: Optional[torch.Tensor] = None,
               scale_factors: Optional[torch.Tensor] = None) -> torch.Tensor:
        A = sum(l.h * l.w for l in levels)
        n = levels[0].n
        if out is None:
            out = torch.empty(n, A, 5 + num_classes, dtype=torch.float32, device=self.device)
            self._keep.append(out)
        arr = (View * len(levels))(*[l.as_c() for l in levels])
        st = (C.c_int32 * len(levels))(*strides) if strides is not None else None
        sf = None
        if scale_factors is not None:
            assert scale_factors.dtype == torch.float32 and scale_factors.is_contiguous() and scale_factors.numel() == 4 * n
            sf = scale_factors.data_ptr()
        check(self.lib.glsdet_yolox_decode(arr, len(levels), num_classes, in_h, in_w, st, mode,
                                           out.data_ptr(), out.numel(), sf, _stream_ptr(self.stream)), "yolox_decode")
        return out

    def nms_buffers(self, n: int, A: int, max_cand: int, max_det: int):
        nbytes = self.lib.glsdet_nms_workspace_bytes(n, A, max_cand)
        return {"ws": self.raw(nbytes), "n": n, "A": A, "max_cand": max_cand, "max_det": max_det,
                "dets": torch.zeros(n, max_det, 7, dtype=torch.float32, device=self.device),
                "count": torch.zeros(2 * n, dtype=torch.int32, device=self.device),
                "status": torch.zeros(1, dtype=torch.int32, device=self.device)}

    def nms(self, pred: torch.Tensor, num_classes: int, box_mode: int, conf_thres: float, nms_thres: float, nb):
        n, A = pred.shape[0], pred.shape[1]
        assert pred.is_contiguous() and pred.dtype == torch.float32 and n == nb["n"] and A == nb["A"]
        check(self.lib.glsdet_nms(pred.data_ptr(), n, A, num_classes, box_mode, conf_thres, nms_thres,
                                  nb["max_cand"], nb["max_det"], nb["dets"].data_ptr(), nb["count"].data_ptr(),
                                  nb["status"].data_ptr(), nb["ws"].data_ptr(), nb["ws"].numel(),
                                  _stream_ptr(self.stream)), "nms")
        return nb["dets"], nb["count"], nb["status"]

    def pack_detections(self, nb, cap: int) -> torch.Tensor:
        """Append the fixed-capacity exchange record of the multi-GPU path to the op sequence:
        nb["packed"] = fp32 [n, cap+1, 7] (glsdet_pack_detections), allocated once here."""
        n = nb["dets"].shape[0]
        if "packed" not in nb or nb["packed"].shape[1] != cap + 1:
            nb["packed"] = torch.zeros(n, cap + 1, 7, dtype=torch.float32, device=self.device)
        check(self.lib.glsdet_pack_detections(nb["dets"].data_ptr(), nb["count"].data_ptr(), n, nb["max_det"], cap,
                                              nb["packed"].data_ptr(), _stream_ptr(self.stream)), "pack_detections")
        return nb["packed"]

    # ------------------------------------------------------------------ ResNet / FPN / GFL / MPHead ops
    def pack_resnet_stem(self, w: torch.Tensor, scale: torch.Tensor, bias: torch.Tensor):
        """conv1.weight [64,3,7,7] -> [64][7][8][4] (tap 7 and channel 3 zero) in the engine dtype, + folded BN"""
        assert tuple(w.shape) == (64, 3, 7, 7)
        wp = torch.zeros(64, 7, 8, 4, dtype=torch.float32)
        wp[:, :, :7, :3] = w.float().permute(0, 2, 3, 1)
        assert wp.numel() == self.lib.glsdet_resnet_stem_weight_elems()
        return (self.upload(wp.to(_TORCH_DT[self.dt])), self.upload(scale.float()), self.upload(bias.float()))

    def resnet_stem(self, img: torch.Tensor, packed, act: str = "relu", out: Optional[TView] = None) -> TView:
        """7x7 stride-2 stem conv + BN + act straight from the fp32 NCHW image (glsdet_resnet_stem)"""
        assert img.dtype == torch.float32 and img.is_contiguous() and img.device.type == "cuda"
        n, cin, H, W = img.shape
        if out is None:
            out = self.tensor(n, (H + 1) // 2, (W + 1) // 2, 64)
        check(self.lib.glsdet_resnet_stem(img.data_ptr(), n, cin, H, W, packed[0].data_ptr(), packed[1].data_ptr(), packed[2].data_ptr(),
                                          ACT[act], C.byref(out.as_c()), _stream_ptr(self.stream)), "resnet_stem")
        return out

    def resnet_stem_pool(self, img: torch.Tensor, packed, out: Optional[TView] = None) -> TView:
        """... + ReLU + MaxPool2d(3, 2, 1) in the same launch (glsdet_resnet_stem_pool)"""
        assert img.dtype == torch.float32 and img.is_contiguous() and img.device.type == "cuda"
        n, cin, H, W = img.shape
        hc, wc = (H + 1) // 2, (W + 1) // 2
        if out is None:
            out = self.tensor(n, (hc + 1) // 2, (wc + 1) // 2, 64)
        check(self.lib.glsdet_resnet_stem_pool(img.data_ptr(), n, cin, H, W, packed[0].data_ptr(), packed[1].data_ptr(),
                                               packed[2].data_ptr(), C.byref(out.as_c()), _stream_ptr(self.stream)), "resnet_stem_pool")
        return out

    def nchw_pack(self, img: torch.Tensor, out: Optional[TView] = None) -> TView:
        assert img.dtype == torch.float32 and img.is_contiguous() and img.device.type == "cuda"
        n, cin, H, W = img.shape
        if out is None:
            out = self.tensor(n, H, W, ceil_to(cin, 8))
        check(self.lib.glsdet_nchw_pack(img.data_ptr(), n, cin, H, W, C.byref(out.as_c()), _stream_ptr(self.stream)),
              "nchw_pack")
        return out

    def pool2d(self, x: TView, k: int, stride: int, pad: int, out: Optional[TView] = None) -> TView:
        if out is None:
            out = self.tensor(x.n, (x.h + 2 * pad - k) // stride + 1, (x.w + 2 * pad - k) // stride + 1, x.c, x.dtype)
        check(self.lib.glsdet_pool2d(C.byref(x.as_c()), C.byref(out.as_c()), k, stride, pad, _stream_ptr(self.stream)),
              "pool2d")
        return out

    def upsample_add(self, coarse: TView, fine: TView) -> TView:
        check(self.lib.glsdet_upsample_add(C.byref(coarse.as_c()), C.byref(fine.as_c()), _stream_ptr(self.stream)),
              "upsample_add")
        return fine

    def groupnorm(self, x: TView, groups: int, gamma: torch.Tensor, beta: torch.Tensor, eps: float, act: str = "relu",
                  out: Optional[TView] = None) -> TView:
        if out is None:
            out = x
        ws = self.raw(self.lib.glsdet_groupnorm_workspace_bytes(x.n, groups))
        check(self.lib.glsdet_groupnorm(C.byref(x.as_c()), C.byref(out.as_c()), groups, gamma.data_ptr(), beta.data_ptr(),
                                        eps, ACT[act], ws.data_ptr(), _stream_ptr(self.stream)), "groupnorm")
        return out

    def groupnorm_multi(self, xs: Sequence[TView], groups: int, gammas, betas, eps: float, act: str = "relu",
                        pre=None) -> List[TView]:
        """In place on each x: up to 16 tensors of the same n / C per launch pair.  pre[i]: the partial sums the conv
        that produced xs[i] already wrote (Engine.conv_gnstats), or None."""
        done = 0
        while done < len(xs):
            part = list(range(done, min(done + 16, len(xs))))
            n = len(part)
            ws = self.raw(n * self.lib.glsdet_groupnorm_workspace_bytes(xs[0].n, groups))
            xa = (View * n)(*[xs[i].as_c() for i in part])
            ga = (C.c_void_p * n)(*[gammas[i].data_ptr() for i in part])
            ba = (C.c_void_p * n)(*[betas[i].data_ptr() for i in part])
            if pre is not None and any(pre[i] is not None for i in part):
                pa = (C.c_void_p * n)(*[(pre[i].data_ptr() if pre[i] is not None else None) for i in part])
                check(self.lib.glsdet_groupnorm_multi_pre(xa, xa, n, groups, ga, ba, eps, ACT[act], ws.data_ptr(), pa,
                                                          _stream_ptr(self.stream)), "groupnorm_multi_pre")
            else:
                check(self.lib.glsdet_groupnorm_multi(xa, xa, n, groups, ga, ba, eps, ACT[act], ws.data_ptr(),
                                                      _stream_ptr(self.stream)), "groupnorm_multi")
            done += n
        return list(xs)

    def conv_gnstats(self, x: TView, packed, pad: int, groups: int, out: Optional[TView] = None):
        """3x3 stride-1 conv (no activation: GroupNorm follows) that also writes the GroupNorm partial sums of its output
        (glsdet_conv2d_gnstats).  -> (y, stats) or (y, None) when no statistics form applies or (autotune) the plain conv
        is faster by more than the statistics pass costs -- y is then produced by Engine.conv."""
        wdev, sdev, bdev, cout, R, S = packed
        if out is None:
            out = self.tensor(x.n, x.h + 2 * pad - R + 1, x.w + 2 * pad - S + 1, cout)
        # Off unless GLSDET_GN_FUSION=1: measured (MPDet, 8 x 800 x 1344, A/B on one box) the statistics pass it saves costs
        # 30 us per GroupNorm launch but the tower convs, MFMA bound at ~1000 TFLOP/s, pay 8 % for the sums in their store
        # phase (148 -> 162 us each): 1214 vs 1230 img/s.  The HBM-bound pass overlaps the other batches' convs; the convs do not.
        if (R, S, pad) != (3, 3, 1) or x.dtype != out.dtype or not os.environ.get("GLSDET_GN_FUSION"):
            return self.conv(x, packed, 1, pad, "none", out=out), None
        stats = self.raw(self.lib.glsdet_conv2d_gnstats_bytes(out.n, out.h, out.w, groups))
        d = ConvDesc()
        d.x, d.y, d.res = x.as_c(), out.as_c(), View()
        d.w, d.scale, d.bias = wdev.data_ptr(), sdev.data_ptr(), bdev.data_ptr()
        d.R, d.S, d.stride, d.pad, d.act, d.tile_hint = 3, 3, 1, 1, ACT["none"], 8
        st = _stream_ptr(self.stream)
        if self.autotune:
            key = ("gnstats", x.n, x.h, x.w, x.c, x.sn, x.sh, x.sw, out.c, out.sn, out.sh, out.sw, groups, out.dtype)
            if key not in self._tuned:
                best, us = C.c_int32(0), C.c_float(0)
                hint = -1
                if self.lib.glsdet_conv2d_gnstats_tune(C.byref(d), groups, stats.data_ptr(), st, C.byref(best), C.byref(us)) == 0:
                    hint = best.value
                    # against the plain conv's best variant + the statistics pass it saves (one read of the output at ~4 TB/s)
                    b1, u1 = C.c_int32(0), C.c_float(0)
                    d.tile_hint = 0
                    saved = out.n * out.h * out.w * out.c * _ESIZE[out.dtype] / 4e6
                    if self.lib.glsdet_conv2d_tune(C.byref(d), st, C.byref(b1), C.byref(u1)) == 0 and us.value > u1.value + saved:
                        hint = -1
                self._tuned[key] = hint
                self._tune_dirty = True
            if self._tuned[key] < 0:
                return self.conv(x, packed, 1, pad, "none", out=out), None
            d.tile_hint = self._tuned[key]
        if self.lib.glsdet_conv2d_gnstats(C.byref(d), groups, stats.data_ptr(), st) != 0:
            return self.conv(x, packed, 1, pad, "none", out=out), None
        return out, stats

    def proxy_scores(self, feat: TView, dots: TView, counts: Sequence[int], gamma: float,
                     out: Optional[TView] = None) -> TView:
        nc = len(counts)
        if out is None:
            out = self.tensor(feat.n, feat.h, feat.w, ceil_to(nc, 8), F32)
        arr = (C.c_int32 * nc)(*counts)
        check(self.lib.glsdet_proxy_scores(C.byref(feat.as_c()), C.byref(dots.as_c()), arr, nc, gamma,
                                           C.byref(out.as_c()), _stream_ptr(self.stream)), "proxy_scores")
        return out

    def gfl_buffers(self, n: int, n_levels: int, max_cand: int, nms_pre: int, max_det: int):
        nbytes = self.lib.glsdet_gfl_workspace_bytes(n, n_levels, max_cand, nms_pre)
        return {"ws": self.raw(nbytes), "n": n, "max_cand": max_cand, "nms_pre": nms_pre, "max_det": max_det,
                "dets": torch.zeros(n, max_det, 7, dtype=torch.float32, device=self.device),
                "count": torch.zeros(2 * n, dtype=torch.int32, device=self.device),
                "status": torch.zeros(1, dtype=torch.int32, device=self.device)}

    def gfl_detect(self, cls: Sequence[TView], reg: Sequence[TView], strides: Sequence[int], num_classes: int,
                   reg_max: int, in_h: int, in_w: int, score_thr: float, iou_thr: float, nb,
                   img_hw: Optional[torch.Tensor] = None, scale_factors: Optional[torch.Tensor] = None):
        L = len(cls)
        ca = (View * L)(*[l.as_c() for l in cls])
        ra = (View * L)(*[l.as_c() for l in reg])
        st = (C.c_int32 * L)(*strides)
        for t, k in ((img_hw, 2), (scale_factors, 4)):
            assert t is None or (t.dtype == torch.float32 and t.is_contiguous() and t.numel() == k * nb["n"])
        check(self.lib.glsdet_gfl_detect(ca, ra, L, st, num_classes, reg_max, in_h, in_w,
                                         img_hw.data_ptr() if img_hw is not None else None,
                                         scale_factors.data_ptr() if scale_factors is not None else None,
                                         score_thr, nb["nms_pre"], iou_thr, nb["max_cand"], nb["max_det"],
                                         nb["dets"].data_ptr(), nb["count"].data_ptr(), nb["status"].data_ptr(),
                                         nb["ws"].data_ptr(), nb["ws"].numel(), _stream_ptr(self.stream)), "gfl_detect")
        return nb["dets"], nb["count"], nb["status"]

    def branch(self, b: int):
        """Ops emitted until the next branch(0) belong to independent branch b (1..8): quadrant
        convs, the l/r/t/b stitch convs, the cls/reg towers.  No-op outside plan recording."""
        # Opt-in (GLSDET_BRANCHES=1).  Measured on MI355X: alone, forking the quadrant / tower
        # convs into parallel graph branches gains ~2 %; with two batches in flight (bench) the
        # fork/join nodes serialise the two graphs against each other and cost 15 % img/s.
        if not os.environ.get("GLSDET_BRANCHES"):
            return
        check(self.lib.glsdet_plan_set_branch(b), "plan_set_branch")

    def save_tune_cache(self):
        if not self._tune_cache_path or not getattr(self, "_tune_dirty", False):
            return
        with _TUNE_IO_LOCK:                         # other lanes' tuners insert into the shared table meanwhile: snapshot it
            data = {}
            if os.path.exists(self._tune_cache_path):
                try:
                    with open(self._tune_cache_path) as f:
                        data = json.load(f)
                except ValueError:                  # another rank's half-written file from before the rename below existed
                    data = {}
            data[self._dtype_name] = {json.dumps(list(k)): v for k, v in list(self._tuned.items())}
            tmp = "%s.%d.%d.tmp" % (self._tune_cache_path, os.getpid(), threading.get_ident())   # whole files only, one tmp per writer
            with open(tmp, "w") as f:
                json.dump(data, f)
            os.replace(tmp, self._tune_cache_path)
            self._tune_dirty = False

    def new_plan(self) -> Plan:
        return Plan(self.lib)


def _fuse_credit_us() -> float:
    """Experiment switch (GLSDET_FUSE_CREDIT_US, default 0): microseconds a fused form is credited per launch it removes when
    the tuner compares it with the separate launches (the graph edge a removed launch no longer pays is not in either timing)."""
    return float(os.environ.get("GLSDET_FUSE_CREDIT_US", "0"))


def fold_bn(gamma, beta, mean, var, eps: float):
    """BatchNorm (eval) -> per-channel scale/bias, computed in fp64."""
    s = gamma.double() / torch.sqrt(var.double() + eps)
    return s.float(), (beta.double() - mean.double() * s).float()
