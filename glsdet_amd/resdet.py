"""ResNet-50 + FPN + GFLHead / MPHead lowered to libglsdet_hip ops (SURVEY section 8a rows
A10, A11) and the plan-backed detector around them.  State-dict keys are mmdet's:

    backbone.{conv1,bn1,layer{1..4}.{i}.{conv1,bn1,conv2,bn2,conv3,bn3,downsample.{0,1}}}
                                              ufp/mmdet/models/backbones/resnet.py:371-646
    neck.{lateral_convs,fpn_convs}.{i}.conv   ufp/mmdet/models/necks/fpn.py:62-148
    bbox_head.{cls_convs,reg_convs}.{i}.{conv,gn}, gfl_cls | gfl_cls_conv + proxies, gfl_reg,
    scales.{l}.scale                          ufp/mmdet/models/dense_heads/{gfl_head,mp_head}.py

Graph-level choices (all numerically the reference's arithmetic):
  * BN folded into the conv epilogue; the Bottleneck's `out += identity; relu` is the
    residual-before-activation epilogue of conv3 (no separate add / ReLU kernels);
  * the first cls and reg tower convs read the same feature -> one GEMM with Cout 512, and one
    GroupNorm over 64 groups of 8 channels (= the two GN32 side by side);
  * `Scale` is folded into the gfl_reg weights per level; F.normalize(proxies) is folded
    into a 1x1 weight, the per-position |feat| and the per-class softmax are one small kernel.
"""
from __future__ import annotations

import threading
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import os

import torch

from . import _lib
from .engine import Engine, F32, TView, fold_bn

# Width of the augmentation of the folded non-local associations ('gram' / 'pair': x' = [x ; 1 ; 0...], "C+a" in the comments
# below).  One channel would do; 64 keeps the contraction length C+a a whole number of the conv kernels' 128-byte K steps
# (their tail-free form).  Measured on config 3 as named (8 x 800 x 1344, one box): a = 8 / 32 / 64 -> 961 / 972 / 982 img/s.
GL_AUG = int(os.environ.get("GLSDET_GL_AUG", "64"))

RESNET_BN_EPS = 1e-5
GN_EPS = 1e-5
STAGE_BLOCKS = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3)}


class GlPending:
    """A GL plug-in level not emitted yet: the FPN decides how (ResDetBuilder.fpn -> gl_lateral / gl_fusion)."""
    __slots__ = ("p", "x", "assoc")

    def __init__(self, p: str, x: TView, assoc: str):
        self.p, self.x, self.assoc = p, x, assoc


class GlDeferred:
    """x + channel_conv(st) of a GL plug-in level whose linear channel_conv has not been applied yet: the only reader is the
    FPN lateral 1x1 conv (no norm, no activation), which takes it in (ResDetBuilder.fpn)."""
    __slots__ = ("p", "x", "st")

    def __init__(self, p: str, x: TView, st: TView):
        self.p, self.x, self.st = p, x, st


class ResDetBuilder:
    def __init__(self, eng: Engine, sd: Dict[str, torch.Tensor]):
        self.e = eng
        self.sd = {k: v.detach().cpu() for k, v in sd.items()}
        self._packed = {}
        self.trace = None       # tests: dict name -> NCHW fp32 snapshot of every stored tensor (eager emission only)

    def _rec(self, name: str, v: TView, c0: int = 0, c1: Optional[int] = None) -> TView:
        """Per-layer trace for the teacher-forced parity test (tests/test_resdet.py): a snapshot of the tensor the op at
        `name` just stored, under the store-point names of oracle/mpdet_oracle.py.  Eager emission only."""
        if self.trace is not None:
            t = v.to_nchw()
            self.trace[name] = t[:, c0:(c1 if c1 is not None else t.shape[1])].cpu()
        return v

    # ------------------------------------------------------------------ weights
    def _pack(self, key, parts, cin_pad):
        k = (key, cin_pad)
        if k not in self._packed:
            self._packed[k] = self.e.pack_conv(parts, cin_pad)
        return self._packed[k]

    def _bn_part(self, conv: str, bn: str):
        s, b = fold_bn(self.sd[bn + ".weight"], self.sd[bn + ".bias"], self.sd[bn + ".running_mean"],
                       self.sd[bn + ".running_var"], RESNET_BN_EPS)
        return self.sd[conv + ".weight"], s, b

    def _plain_part(self, p: str, mul: float = 1.0):
        w = self.sd[p + ".weight"].float() * mul
        b = self.sd.get(p + ".bias")
        b = torch.zeros(w.shape[0]) if b is None else b.float() * mul
        return w, torch.ones(w.shape[0]), b

    def _dev(self, key: str, t: torch.Tensor) -> torch.Tensor:
        if key not in self._packed:
            self._packed[key] = self.e.upload(t.float().contiguous())
        return self._packed[key]

    # ------------------------------------------------------------------ backbone
    def bottleneck(self, p: str, x: TView, stride: int, cat: Optional[TView] = None) -> TView:
        """resnet.py:263-303 (style='pytorch': the stride sits on the 3x3).
        cat: a [n, h, w, mid + cin] buffer whose channels [mid, mid + cin) ARE x (the caller had x's producer write there);
        stride 1 with a downsample branch only.  conv2 then writes channels [0, mid) and conv3 and the downsample conv are ONE
        1x1 over the concatenation,  relu(s3 (W3 h) + b3 + sd (Wd x) + bd) = relu([s3 W3 | sd Wd] [h ; x] + b3 + bd):
        the identity tensor (275 MB at layer1 of the benchmark) is neither written nor read back.  BN scales ride in the
        weights (float64 on the host), so the fp16 rounding of the composed weights differs from the reference's by what a
        folded BN always costs; a trace keeps the reference's op sequence."""
        e = self.e
        if cat is not None:
            assert stride == 1 and p + ".downsample.0.weight" in self.sd
            w1 = self.sd[p + ".conv1.weight"]
            mid, cin = w1.shape[0], w1.shape[1]
            assert cat.c == mid + cin and x.c == cin
            t = e.conv(x, self._pack(p + ".conv1", [self._bn_part(p + ".conv1", p + ".bn1")], x.c), 1, 0, "relu")
            e.conv(t, self._pack(p + ".conv2", [self._bn_part(p + ".conv2", p + ".bn2")], t.c), 1, 1, "relu", out=cat.channels(0, mid))
            key = (p + ".conv3+downsample", cat.c)
            if key not in self._packed:
                w3, s3, b3 = self._bn_part(p + ".conv3", p + ".bn3")
                wd, sd_, bd = self._bn_part(p + ".downsample.0", p + ".downsample.1")
                w = torch.cat([w3.double() * s3.double().view(-1, 1, 1, 1), wd.double() * sd_.double().view(-1, 1, 1, 1)], 1).float()
                self._packed[key] = e.pack_conv([(w, torch.ones(w.shape[0]), (b3.double() + bd.double()).float())], cat.c)
            return e.conv(cat, self._packed[key], 1, 0, "relu")
        t = self._rec(p + ".conv1", e.conv(x, self._pack(p + ".conv1", [self._bn_part(p + ".conv1", p + ".bn1")], x.c), 1, 0, "relu"))
        t = self._rec(p + ".conv2", e.conv(t, self._pack(p + ".conv2", [self._bn_part(p + ".conv2", p + ".bn2")], t.c), stride, 1, "relu"))
        idn = x
        if p + ".downsample.0.weight" in self.sd:
            idn = e.conv(x, self._pack(p + ".downsample", [self._bn_part(p + ".downsample.0", p + ".downsample.1")], x.c),
                         stride, 0, "none")
            self._rec(p + ".downsample", idn)
        return self._rec(p + ".conv3", e.conv(t, self._pack(p + ".conv3", [self._bn_part(p + ".conv3", p + ".bn3")], t.c), 1, 0, "relu",
                                              res=idn, res_first=True))

    def resnet(self, p: str, img: torch.Tensor, depth: int = 50, out_indices: Sequence[int] = (0, 1, 2, 3)) -> List[TView]:
        """resnet.py:631-646."""
        e = self.e
        w1 = self.sd[p + ".conv1.weight"]
        cat10 = None
        if tuple(w1.shape) == (64, 3, 7, 7) and img.shape[1] == 3 and not os.environ.get("GLSDET_NO_RSTEM_FUSION"):
            # stem conv straight from the fp32 NCHW image (glsdet_resnet_stem): no packed image, K = 7 x 8 x 4 instead of 7 x 7 x 8
            key = (p + ".conv1", "rstem")
            if key not in self._packed:
                self._packed[key] = e.pack_resnet_stem(*self._bn_part(p + ".conv1", p + ".bn1"))
            if not os.environ.get("GLSDET_NO_RSTEM_POOL"):
                # layer1.0's identity branch composed with its conv3 (bottleneck(cat=)): the pooled stem output is written into
                # the upper channels of the buffer conv3 reads
                p10 = "%s.layer1.0" % p
                if self.trace is None and not os.environ.get("GLSDET_NO_DOWNSAMPLE_FOLD") and p10 + ".downsample.0.weight" in self.sd \
                        and self.sd[p10 + ".conv2.weight"].shape[-1] == 3:
                    mid = self.sd[p10 + ".conv1.weight"].shape[0]
                    n_, H_, W_ = img.shape[0], img.shape[2], img.shape[3]
                    hc, wc = (H_ + 1) // 2, (W_ + 1) // 2
                    cat10 = e.tensor(n_, (hc + 1) // 2, (wc + 1) // 2, mid + 64)
                    x = e.resnet_stem_pool(img, self._packed[key], out=cat10.channels(mid, mid + 64))
                else:
                    x = e.resnet_stem_pool(img, self._packed[key])        # ... and the max pool in its epilogue
            else:
                x = e.pool2d(e.resnet_stem(img, self._packed[key], "relu"), 3, 2, 1)
        else:
            x = e.nchw_pack(img)
            x = e.conv(x, self._pack(p + ".conv1", [self._bn_part(p + ".conv1", p + ".bn1")], x.c), 2, 3, "relu")
            x = e.pool2d(x, 3, 2, 1)
        self._rec(p + ".maxpool", x)
        outs = []
        for i, nblocks in enumerate(STAGE_BLOCKS[depth]):
            for j in range(nblocks):
                x = self.bottleneck("%s.layer%d.%d" % (p, i + 1, j), x, 2 if (j == 0 and i > 0) else 1,
                                    cat=cat10 if (i == 0 and j == 0) else None)
            if i in out_indices:
                outs.append(x)
        return outs

    # ------------------------------------------------------------------ GL-fusion plug-in (BASELINE config 3)
    def _wmat(self, key: str, w: torch.Tensor) -> TView:
        """A conv weight [rows, cols] as an Engine.matrix (so that it can be the ACTIVATION operand of a GEMM)."""
        if key not in self._packed:
            m = self.e.matrix(w.shape[0], w.shape[1])
            pitch = m.sw
            t = torch.zeros(w.shape[0], pitch, dtype=torch.float32)
            t[:, : w.shape[1]] = w.float()
            m.buf.view(torch.float16 if m.dtype == 0 else torch.float32)[: t.numel()].copy_(t.flatten())
            self._packed[key] = m
        return self._packed[key]

    @staticmethod
    def gl_assoc(assoc: str, ci: int, C_: int, n: int, R: Optional[int] = None) -> str:
        """The association nonlocal_gemm uses for quadrants of n pixels ('auto': the fewest multiplies; GLSDET_GL_ASSOC
        overrides).  R: width of the linear maps taken in behind the block (gl_lateral: the FPN width), None = none."""
        if assoc != "auto":
            return assoc
        Ca = C_ + GL_AUG
        Ro = C_ if R is None else R
        cost = {"gram": Ca * Ca * n + Ro * Ca * Ca + Ro * C_ * Ca + n * C_ * Ro,
                "pair": n * Ca * Ca + n * n * Ca + Ro * n * Ca + n * n * Ro + (n * C_ * Ro if R is not None else 0)}
        if R is None:
            cost.update({"re": 4 * ci * C_ * n + ci * ci * n + C_ * ci * ci, "dir": 2 * n * n * ci + 3 * n * ci * C_})
        if Ro == C_ and n < 2 * C_:
            # at full width 'gram' pays two C^3 products and a bias product per window: by the multiply count it ties with
            # 'pair' around n = C, measured (config 3 as named, 50 x 84 level: C = 1024, n = 1050) it is 1006 us against 927
            del cost["gram"]
        return os.environ.get("GLSDET_GL_ASSOC") or min(cost, key=cost.get)

    def nonlocal_gemm(self, ps: Sequence[str], xs: Sequence[TView], outs: Sequence[TView], assoc: str = "auto", post=None):
        """Non_local_Block (drone/models/new/Non_local_family.py:6-50) at ResNet widths (C = 512...2048), for the
        quadrants `xs` (views with the batch in n) of ONE feature map, written to `outs`:
            out = x + conv_out( (theta^T phi / N) g^T )            (dot product, divide by N, no softmax)
        Every product is a 1x1 "conv" on the MFMA conv kernels whose weight operand is an activation matrix of the
        same image (Engine.matrix / as_weight), one problem per (image, quadrant), eight problems per launch.
        Three associations, chosen per map by their multiply count (N = pixels of a quadrant, C = channels, ci = inner):
          're'   : Phi^T|G^T = [Wphi;Wg] Xq^T ; M = Phi^T G / N ; P = Wout M^T ; out = x + Theta P^T + b   (4 ci C N + ci^2 N + C ci^2)
          'dir'  : S = Theta Phi^T / N ; G^T = Wg Xq^T ; Y = S G ; out = x + Y Wout^T + b                   (2 N^2 ci + 3 N ci C)
          'gram' : with x' = [x ; 1] (so that every bias is a matrix column):  G' = X'^T X' / N ;
                   Q' = (Wout Wg') G' (Wphi'^T Wtheta') ; out = x + X' Q'^T + b                             (2 C^2 N + 2 C^3)
                   -- the two weight products are constants folded at build time, so the only per-pixel work left is the
                   Gram matrix of the quadrant and one C x C product per pixel (no theta / phi / g projections at all).
          'pair' : S = X' (Wtheta'^T Wphi') X'^T / N  (the N x N pair matrix straight from the window) ;
                   out = x + S (X' (Wout Wg')^T) + b                                                        (2 N C^2 + 2 N^2 C)
                   -- 'dir' with theta/phi and g/conv_out folded pairwise at build time: four products instead of six.
        (the reference order is 'dir' with fp32 intermediates; here S / Y / M / P / G' / Q' / U / V are stored in the engine
        dtype)."""
        e = self.e
        ci = self.sd[ps[0] + ".theta.weight"].shape[0]
        C_ = xs[0].c
        assert self.sd[ps[0] + ".conv_out.weight"].shape[0] == C_
        nimg = xs[0].n
        Ns = [x.h * x.w for x in xs]
        Np = (max(Ns) + 7) // 8 * 8
        assoc = self.gl_assoc(assoc, ci, C_, max(Ns))
        assert post is None or assoc == "gram" or (assoc == "pair" and not post["res_x"]), \
            "linear maps behind the block ride in the per-window matrices of 'gram' (any) and 'pair' (to another width) only"
        w2 = lambda p, nm: self.sd["%s.%s.weight" % (p, nm)].float().reshape(self.sd["%s.%s.weight" % (p, nm)].shape[0], -1)
        bias = lambda p, nm: self.sd["%s.%s.bias" % (p, nm)]
        pk = lambda p, nm: self._pack("%s.%s" % (p, nm), [self._plain_part("%s.%s" % (p, nm))], C_)

        def bias_rows(key, b, cols, valid):       # [rows, cols] matrix holding b[r] in the first `valid` columns
            k = ("biasrows", key, cols, valid)
            if k not in self._packed:
                m = e.matrix(b.numel(), cols)
                t = torch.zeros(b.numel(), m.sw)
                t[:, :valid] = b.float()[:, None]
                m.buf.view(torch.float16 if m.dtype == 0 else torch.float32)[: t.numel()].copy_(t.flatten())
                self._packed[k] = m
            return self._packed[k]

        jobs = [(b, q) for b in range(nimg) for q in range(len(xs))]
        xq = {(b, q): xs[q].image(b) for b, q in jobs}
        oq = {(b, q): outs[q].image(b) for b, q in jobs}
        if assoc in ("gram", "pair"):            # the pixel count is a contraction length there: whole 64-byte K steps as well
            np32 = (max(Ns) + 31) // 32 * 32          # (64 measured the same)
            if assoc == "gram":
                return self._nonlocal_gram(ps, xs, outs, jobs, xq, oq, Ns, np32, post)
            return self._nonlocal_pair(ps, xs, outs, jobs, xq, oq, Ns, np32, post)
        # theta for every (image, quadrant): a plain conv per quadrant over the whole batch
        theta = [e.tensor(x.n, x.h, x.w, ci) for x in xs]
        e.conv_many(list(xs), [pk(p, "theta") for p in ps], 1, 0, "none", theta)
        th = {(b, q): theta[q].image(b) for b, q in jobs}
        # dense copy of each quadrant: rows = pixels (the weight operand of the transposed projections)
        Xq, dense = {}, []
        for (b, q) in jobs:
            m = e.matrix(Np, C_)
            Xq[b, q] = m
            x = xs[q]
            dense.append(TView(m.buf, 0, 1, x.h, x.w, C_, x.h * x.w * m.sw, x.w * m.sw, m.sw, m.dtype))
        e.copy_many([xq[j] for j in jobs], dense)            # one launch per 32 (image, quadrant) pairs
        if assoc == "re":
            # [Phi^T ; G^T] (2ci x Np) = [Wphi ; Wg] (as pixels) x Xq^T, bias per row through the residual operand
            PG = {j: e.matrix(2 * ci, Np) for j in jobs}
            srcs, packs, ress = [], [], []
            for (b, q) in jobs:
                p = ps[q]
                srcs.append(self._wmat(p + ".phig", torch.cat([w2(p, "phi"), w2(p, "g")], 0)))
                packs.append(e.as_weight(Xq[b, q]))
                ress.append(bias_rows(p + ".phig", torch.cat([bias(p, "phi").float(), bias(p, "g").float()]), Np, Ns[q]))
            e.conv_many(srcs, packs, 1, 0, "none", [PG[j] for j in jobs], ress)
            rows = lambda m, r0, r1: TView(m.buf, m.off + r0 * m.sw, 1, 1, r1 - r0, m.c, (r1 - r0) * m.sw, (r1 - r0) * m.sw, m.sw, m.dtype)
            # M/N (ci x ci): rows a = Phi^T[a][:], weight rows b = G^T[b][:]
            M = {j: e.matrix(ci, ci) for j in jobs}
            e.conv_many([rows(PG[j], 0, ci) for j in jobs],
                        [e.as_weight(rows(PG[b, q], ci, 2 * ci), alpha=1.0 / Ns[q]) for (b, q) in jobs], 1, 0, "none",
                        [M[j] for j in jobs])
            # P (C x ci) = Wout (as pixels) x M^T
            P = {j: e.matrix(C_, ci) for j in jobs}
            e.conv_many([self._wmat(ps[q] + ".wout", w2(ps[q], "conv_out")) for (b, q) in jobs],
                        [e.as_weight(M[j]) for j in jobs], 1, 0, "none", [P[j] for j in jobs])
            # out = x + Theta P^T + bout
            e.conv_many([th[j] for j in jobs], [e.as_weight(P[b, q], bias=bias(ps[q], "conv_out")) for (b, q) in jobs],
                        1, 0, "none", [oq[j] for j in jobs], [xq[j] for j in jobs])
            return list(outs)
        # ---- direct association
        phi = [e.matrix(Np, ci) for _ in jobs]
        # Phi (Np x ci) per job as a weight matrix: conv of the dense quadrant copy (rows beyond N stay zero)
        e.conv_many([TView(Xq[j].buf, 0, 1, 1, Ns[j[1]], C_, Ns[j[1]] * Xq[j].sw, Ns[j[1]] * Xq[j].sw, Xq[j].sw, Xq[j].dtype) for j in jobs],
                    [pk(ps[q], "phi") for (b, q) in jobs], 1, 0, "none",
                    [TView(m.buf, 0, 1, 1, Ns[j[1]], ci, Ns[j[1]] * m.sw, Ns[j[1]] * m.sw, m.sw, m.dtype) for m, j in zip(phi, jobs)])
        # S (N x Np) = Theta Phi^T / N
        S = [e.matrix(Ns[q], Np) for (b, q) in jobs]
        e.conv_many([th[j] for j in jobs],
                    [e.as_weight(m, alpha=1.0 / Ns[j[1]]) for m, j in zip(phi, jobs)], 1, 0, "none",
                    [TView(m.buf, 0, 1, th[j].h, th[j].w, m.c, th[j].h * th[j].w * m.sw, th[j].w * m.sw, m.sw, m.dtype) for m, j in zip(S, jobs)])
        # G^T (ci x Np) = Wg (as pixels) x Xq^T + bg per row
        GT = [e.matrix(ci, Np) for _ in jobs]
        e.conv_many([self._wmat(ps[q] + ".wg", w2(ps[q], "g")) for (b, q) in jobs], [e.as_weight(Xq[j]) for j in jobs], 1, 0, "none",
                    GT, [bias_rows(ps[q] + ".g", bias(ps[q], "g"), Np, Ns[q]) for (b, q) in jobs])
        # Y (N x ci) = S G
        Y = [e.matrix(Ns[q], ci) for (b, q) in jobs]
        e.conv_many(S, [e.as_weight(m) for m in GT], 1, 0, "none", Y)
        # out = x + Y Wout^T + bout, written as an NHWC quadrant
        yv = [TView(m.buf, 0, 1, th[j].h, th[j].w, m.c, th[j].h * th[j].w * m.sw, th[j].w * m.sw, m.sw, m.dtype) for m, j in zip(Y, jobs)]
        e.conv_many(yv, [pk(ps[q], "conv_out") for (b, q) in jobs], 1, 0, "none", [oq[j] for j in jobs], [xq[j] for j in jobs])
        return list(outs)

    def _gl_aug(self, p: str, nm: str, C_: int) -> torch.Tensor:
        """[W | b | 0...]  (rows x C+a, float64) of the 1x1 conv p.nm."""
        w = self.sd["%s.%s.weight" % (p, nm)].double().reshape(self.sd["%s.%s.weight" % (p, nm)].shape[0], -1)
        out = torch.zeros(w.shape[0], C_ + GL_AUG, dtype=torch.float64)
        out[:, :C_] = w
        out[:, C_] = self.sd["%s.%s.bias" % (p, nm)].double()
        return out

    def _gl_post(self, p: str, ps, x: TView, wins, lateral: Optional[str] = None):
        """The linear maps behind the four non-local blocks of plug-in p, composed on the host (float64):
        gl_fusion's own linear channel_conv and residual,  z = x + Wc (x + NL(x)) + bc,  and -- lateral = name of the FPN
        lateral conv that alone reads z -- that conv as well,  Wl z + bl.  With NL(x) = (per-window linear map of x') + bout:
            result = Wr x + Wp NL0(x) + bias_q          (NL0 = NL without bout; Wr x includes the residuals)
            channel_conv only:  Wp = Wc,     Wr = I + Wc       (applied as `+ x` and Wc: res_x),  bias_q = Wc bout_q + bc
            with the lateral:   Wp = Wl Wc,  Wr = Wl (I + Wc),                                    bias_q = Wl (Wc bout_q + bc) + bl"""
        C_ = x.c
        Wc = self.sd[p + ".channel_conv.weight"].double().reshape(C_, -1)
        bc = self.sd.get(p + ".channel_conv.bias")
        bc = torch.zeros(C_, dtype=torch.float64) if bc is None else bc.double()
        bout = lambda q: self.sd[ps[q] + ".conv_out.bias"].double()
        if lateral is None:
            return dict(key=p + "@cc", Wp=Wc, Wr=Wc, res_x=True, bias=lambda q: Wc @ bout(q) + bc, x=x, wins=wins)
        wl, _, bl = self._plain_part(lateral)
        Wl = wl.double().reshape(wl.shape[0], -1)
        return dict(key=p + "@" + lateral, Wp=Wl @ Wc, Wr=Wl + Wl @ Wc, res_x=False,
                    bias=lambda q: Wl @ (Wc @ bout(q) + bc) + bl.double(), x=x, wins=wins)

    def _gl_consts(self, p: str, C_: int):
        """Build-time constants of the folded associations of one non-local block (float64 on the host), with
        W' = [W | b | 0...] (ci x C+a):  A = Wout Wg' (C x C+a) as an activation matrix;  B = Wphi'^T Wtheta' ((C+a) x (C+a))
        as conv weights in both orientations.  -> (A matrix, pack of B^T [out = T B], pack of B [out = X' (Wtheta'^T Wphi')],
        B^T as an activation matrix, pack of the first C rows of B^T [out = T B[:, :C]])"""
        e = self.e
        Ca = C_ + GL_AUG
        k = ("glconst", p)
        if k not in self._packed:
            aug = lambda nm: self._gl_aug(p, nm, C_)
            wout = self.sd[p + ".conv_out.weight"].double().reshape(C_, -1)
            A = (wout @ aug("g")).float()                                            # C x Ca
            B = (aug("phi").t() @ aug("theta")).float()                              # Ca x Ca
            one, zero = torch.ones(Ca), torch.zeros(Ca)
            Bt = B.t().contiguous()
            self._packed[k] = (self._wmat(p + ".glA", A),
                               e.pack_conv([(Bt.reshape(Ca, Ca, 1, 1), one, zero)], Ca),
                               e.pack_conv([(B.contiguous().reshape(Ca, Ca, 1, 1), one, zero)], Ca),
                               self._wmat(p + ".glBt", Bt),                                      # B^T as an activation matrix
                               e.pack_conv([(Bt[:C_].reshape(C_, Ca, 1, 1), torch.ones(C_), torch.zeros(C_))], Ca))   # Q'[:, :C] = T B[:, :C]
        return self._packed[k]

    def _gl_dense(self, xs, jobs, xq, Ns, Np):
        """Row-major dense copies X' [Np x C+a] of the (image, quadrant) windows, channel C preset to one on the valid rows."""
        e = self.e
        C_ = xs[0].c
        tdt = torch.float16 if e.dt == 0 else torch.float32
        Xr, dense = {}, []
        for (b, q) in jobs:
            x, n = xs[q], Ns[q]
            m = e.matrix(Np, C_ + GL_AUG)
            m.buf.view(tdt)[: Np * m.sw].view(Np, m.sw)[:n, C_] = 1.0
            Xr[b, q] = m
            dense.append(TView(m.buf, 0, 1, x.h, x.w, C_, x.h * x.w * m.sw, x.w * m.sw, m.sw, m.dtype))
        e.copy_many([xq[j] for j in jobs], dense)
        return Xr

    def _nonlocal_pair(self, ps, xs, outs, jobs, xq, oq, Ns, Np, post=None):
        """The 'pair' association of nonlocal_gemm (see there).  Per (image, quadrant), X' = the window with a ones channel:
            U = X' (Wtheta'^T Wphi')   (N x C+a)        S = U X'^T / N   (N x N: theta_i . phi_j / N)
            V^T = (Wout Wg') X'^T      (C x N)          out = x + S V + bout
        post (see _gl_post; to another width R only): the linear maps behind the block applied to V instead of to the pixels,
            V2^T = (Wp Wout Wg') X'^T  (R x N)          out = Wr x + S V2 + bias_q     (Wr x: ONE 1x1 conv over the whole map)."""
        e = self.e
        C_ = xs[0].c
        Ca = C_ + GL_AUG
        Xr = self._gl_dense(xs, jobs, xq, Ns, Np)
        act = lambda m, n: TView(m.buf, 0, 1, 1, n, m.c, n * m.sw, n * m.sw, m.sw, m.dtype)       # the first n rows as pixels
        U = {j: e.matrix(Np, Ca) for j in jobs}
        e.conv_many([act(Xr[j], Ns[j[1]]) for j in jobs], [self._gl_consts(ps[q], C_)[2] for (b, q) in jobs], 1, 0, "none",
                    [act(U[j], Ns[j[1]]) for j in jobs])
        S = {(b, q): e.matrix(Ns[q], Np) for (b, q) in jobs}
        e.conv_many([act(U[j], Ns[j[1]]) for j in jobs], [e.as_weight(Xr[b, q], alpha=1.0 / Ns[q]) for (b, q) in jobs], 1, 0, "none",
                    [S[j] for j in jobs])
        sv = [TView(S[b, q].buf, 0, 1, xs[q].h, xs[q].w, S[b, q].c, xs[q].h * xs[q].w * S[b, q].sw, xs[q].w * S[b, q].sw, S[b, q].sw,
                    S[b, q].dtype) for (b, q) in jobs]
        if post is None:
            VT = {j: e.matrix(C_, Np) for j in jobs}
            e.conv_many([self._gl_consts(ps[q], C_)[0] for (b, q) in jobs], [e.as_weight(Xr[j]) for j in jobs], 1, 0, "none",
                        [VT[j] for j in jobs])
            e.conv_many(sv, [e.as_weight(VT[b, q], bias=self.sd[ps[q] + ".conv_out.bias"]) for (b, q) in jobs], 1, 0, "none",
                        [oq[j] for j in jobs], [xq[j] for j in jobs])
            return list(outs)
        R = post["Wp"].shape[0]
        one, zero = torch.ones(R), torch.zeros(R)
        # Wr x over the whole map (the quadrant windows of t are the residual operands below)
        t = e.conv(post["x"], self._pack(post["key"] + "@x", [(post["Wr"].float().reshape(R, C_, 1, 1), one, zero)], C_), 1, 0, "none")
        tq = post["wins"](t)
        A2 = {}
        for q, p in enumerate(ps):
            wout = self.sd[p + ".conv_out.weight"].double().reshape(C_, -1)
            A2[q] = self._wmat("%s.A2.%d" % (post["key"], q), (post["Wp"] @ wout @ self._gl_aug(p, "g", C_)).float())     # R x Ca
        VT = {j: e.matrix(R, Np) for j in jobs}
        e.conv_many([A2[q] for (b, q) in jobs], [e.as_weight(Xr[j]) for j in jobs], 1, 0, "none", [VT[j] for j in jobs])
        bq = {q: post["bias"](q).float() for q in range(len(ps))}          # (as_weight keeps them alive: it keys a bias by address)
        e.conv_many(sv, [e.as_weight(VT[b, q], bias=bq[q]) for (b, q) in jobs], 1, 0, "none",
                    [oq[j] for j in jobs], [tq[q].image(b) for (b, q) in jobs])
        return list(outs)

    def _nonlocal_gram(self, ps, xs, outs, jobs, xq, oq, Ns, Np, post=None):
        """The 'gram' association of nonlocal_gemm (see there).  Per (image, quadrant): the transposed copy X'^T [C+a x N] of the
        window (the constant-one channel C preset at build time), then
            G' = X'^T X' / N  ((C+a) x (C+a), contraction over the pixels),   T = A G'  (C x C+a),
            Qm = T B[:, :C]  (C x C),   d = T B[:, C] + bout  (fp32 vector),   out = x + x Qm^T + d
        with the constants A = Wout [Wg | bg] (C x C+a) and B = [Wphi | bphi]^T [Wtheta | btheta] ((C+a) x (C+a)) folded
        on the host in float64: the per-pixel product contracts over exactly C channels of the window where it lies (no
        row-major copy), its bias is a vector a 1-pixel product of the plan writes.
        post (see _gl_post): the linear maps behind the block taken into the per-window matrices from the front,
            T2 = (Wp A) G'  (R x C+a),   Qm2 = T2 B[:, :C] + Wr,   d2 = T2 B[:, C] + bias_q,   out = [x +] x Qm2^T + d2
        (R = C or the FPN width): the same three small products, at the OUTPUT width, instead of C x C (and R x C) products per
        pixel of the map."""
        e = self.e
        C_ = xs[0].c
        Ca = C_ + GL_AUG
        tdt = torch.float16 if e.dt == 0 else torch.float32
        rows = lambda m, r0, r1: TView(m.buf, m.off + r0 * m.sw, 1, 1, r1 - r0, m.c, (r1 - r0) * m.sw, (r1 - r0) * m.sw, m.sw, m.dtype)
        consts = lambda p: self._gl_consts(p, C_)
        bout = lambda p: self.sd[p + ".conv_out.bias"]
        Xt = {}
        for (b, q) in jobs:
            t = e.matrix(Ca, Np)
            t.buf.view(tdt)[C_ * t.sw: C_ * t.sw + Ns[q]] = 1.0
            Xt[b, q] = t
        e.transpose_many([xq[j] for j in jobs], [Xt[j] for j in jobs])
        G = {j: e.matrix(Ca, Ca) for j in jobs}
        e.conv_many([Xt[j] for j in jobs], [e.as_weight(Xt[b, q], alpha=1.0 / Ns[q]) for (b, q) in jobs], 1, 0, "none",
                    [G[j] for j in jobs])
        R = C_ if post is None else post["Wp"].shape[0]
        if post is not None:
            # the linear maps behind the block go in FRONT: A2 = Wp A (R x C+a, a constant), so T2 = A2 G' is already at the
            # output width, Qm2 = T2 B[:, :C] + Wr and d2 = T2 B[:, C] + bias_q -- the same three products as without them
            A2 = {}
            for q, p in enumerate(ps):
                wout = self.sd[p + ".conv_out.weight"].double().reshape(C_, -1)
                A2[q] = self._wmat("%s.gA2.%d" % (post["key"], q), (post["Wp"] @ wout @ self._gl_aug(p, "g", C_)).float())
            WrM = self._wmat(post["key"] + ".WrM", post["Wr"].float())
            bq = {q: post["bias"](q).float() for q in range(len(ps))}
        T = {j: e.matrix(R, Ca) for j in jobs}
        e.conv_many([(consts(ps[q])[0] if post is None else A2[q]) for (b, q) in jobs], [e.as_weight(G[j]) for j in jobs], 1, 0, "none",
                    [T[j] for j in jobs])
        Qm = {j: e.matrix(R, C_) for j in jobs}
        d = {j: e.bias_vector(R) for j in jobs}
        e.conv_many([T[j] for j in jobs], [consts(ps[q])[4] for (b, q) in jobs], 1, 0, "none", [Qm[j] for j in jobs],
                    None if post is None else [WrM for _ in jobs])
        e.conv_many([rows(consts(ps[q])[3], C_, C_ + 1) for (b, q) in jobs],
                    [e.as_weight(T[b, q], bias=(bout(ps[q]) if post is None else bq[q])) for (b, q) in jobs], 1, 0, "none",
                    [d[j] for j in jobs])
        res_x = post is None or post["res_x"]
        e.conv_many([xq[j] for j in jobs], [e.as_weight(Qm[j], bias_dev=d[j]) for j in jobs], 1, 0, "none",
                    [oq[j] for j in jobs], [xq[j] for j in jobs] if res_x else None)
        return list(outs)

    def gl_fusion(self, p: str, x: TView, assoc: str = "auto", fold="auto", defer: bool = False):
        """x + Patch_Conv_NonLocal_new(x)  (drone/models/new/yolox10.py:262-266 applied to a ResNet stage output;
        Non_local_family.py:208-252): quadrant non-local blocks at the input resolution, re-stitch (free: the
        blocks write their windows of one buffer), channel_conv, residual."""
        e = self.e
        hh, hw = x.h // 2, x.w // 2
        st = e.tensor(x.n, x.h, x.w, x.c)
        wins = lambda t: [t.window(0, hh, 0, hw), t.window(hh, t.h, 0, hw), t.window(0, hh, hw, t.w), t.window(hh, t.h, hw, t.w)]
        names = ["%s.feat_patchconv_%s_nonlocal" % (p, q) for q in ("lt", "lb", "rt", "rb")]
        # the linear channel_conv (+ x) rides in the per-window matrices of the 'gram' association where a C^3 product per
        # window is cheaper than a C x C product per pixel (fold: True / False / "auto")
        n_max = (x.h - hh) * (x.w - hw)
        a = self.gl_assoc(assoc, self.sd[names[0] + ".theta.weight"].shape[0], x.c, n_max)
        linear = p + ".channel_conv.weight" in self.sd
        if a == "gram" and linear and (fold is True or (fold == "auto" and n_max >= 2 * x.c)):
            self.nonlocal_gemm(names, wins(x), wins(st), "gram", post=self._gl_post(p, names, x, wins))
            return st
        self.nonlocal_gemm(names, wins(x), wins(st), a)
        if defer and linear:                 # the caller is the FPN: its lateral conv composes with the channel_conv
            return GlDeferred(p, x, st)
        if p + ".channel_conv.weight" in self.sd:                       # channel_cat == 'linear': 1x1 conv + bias
            pkc = self._pack(p + ".channel_conv", [self._plain_part(p + ".channel_conv")], st.c)
            return e.conv(st, pkc, 1, 0, "none", res=x)
        s, b = fold_bn(self.sd[p + ".channel_conv.bn.weight"], self.sd[p + ".channel_conv.bn.bias"],
                       self.sd[p + ".channel_conv.bn.running_mean"], self.sd[p + ".channel_conv.bn.running_var"], 1e-3)
        pkc = self._pack(p + ".channel_conv", [(self.sd[p + ".channel_conv.conv.weight"], s, b)], st.c)
        return e.conv(st, pkc, 1, 1, "silu", res=x)                     # BaseConv 3x3 + BN(1e-3) + SiLU, then + x

    def gl_lateral(self, p: str, x: TView, lateral: str, assoc: str = "auto") -> TView:
        """lateral_conv(x + Patch_Conv_NonLocal_new(x)) for a plug-in level whose output only the FPN lateral conv reads
        (GLFusionFPN), linear channel_conv: the whole tail -- conv_out bias, channel_conv, both residuals, the lateral conv --
        rides in the per-window matrices, and the level's first stored tensor after the backbone is the lateral itself at the
        FPN width.  'gram': one C -> F product per pixel; 'pair': V at width F instead of C (F / C of its multiplies and of
        the S V product) and ONE C -> F conv of x over the map.  'dir' / 're': the stored form, then _lateral_of_deferred."""
        e = self.e
        hh, hw = x.h // 2, x.w // 2
        wins = lambda t: [t.window(0, hh, 0, hw), t.window(hh, t.h, 0, hw), t.window(0, hh, hw, t.w), t.window(hh, t.h, hw, t.w)]
        names = ["%s.feat_patchconv_%s_nonlocal" % (p, q) for q in ("lt", "lb", "rt", "rb")]
        a = self.gl_assoc(assoc, self.sd[names[0] + ".theta.weight"].shape[0], x.c, (x.h - hh) * (x.w - hw),
                          R=self.sd[lateral + ".weight"].shape[0])
        if a not in ("gram", "pair"):
            return self._lateral_of_deferred(lateral, self.gl_fusion(p, x, a, fold=False, defer=True))
        out = e.tensor(x.n, x.h, x.w, self.sd[lateral + ".weight"].shape[0])
        self.nonlocal_gemm(names, wins(x), wins(out), a, post=self._gl_post(p, names, x, wins, lateral))
        return out

    def gl_materialise(self, d: GlDeferred) -> TView:
        """x + channel_conv(st) as a stored tensor (gl_fusion's own last step)."""
        pkc = self._pack(d.p + ".channel_conv", [self._plain_part(d.p + ".channel_conv")], d.st.c)
        return self.e.conv(d.st, pkc, 1, 0, "none", res=d.x)

    def _lateral_of_deferred(self, name: str, d: GlDeferred) -> TView:
        """lateral(x + Wc st + bc) = Wl x + (Wl Wc) st + (Wl bc + bl): two 1x1 convs to the FPN width instead of a C x C
        product per pixel followed by one (fpn.py:165-168 on Non_local_family.py:247-250; nothing non-linear lies between
        the two and the lateral conv is the plug-in output's only reader).  At C = 1024 / 2048 that is 0.5 / 1.0 instead of
        1.3 / 4.7 MMAC per pixel."""
        e = self.e
        wl, _, bl = self._plain_part(name)
        C_ = d.x.c
        Wl = wl.double().reshape(wl.shape[0], -1)
        Wc = self.sd[d.p + ".channel_conv.weight"].double().reshape(C_, -1)
        bc = self.sd.get(d.p + ".channel_conv.bias")
        bc = torch.zeros(C_, dtype=torch.float64) if bc is None else bc.double()
        one, zero = torch.ones(wl.shape[0]), torch.zeros(wl.shape[0])
        t = e.conv(d.x, self._pack(name + "@x", [(wl.float(), one, (bl.double() + Wl @ bc).float())], C_), 1, 0, "none")
        w2 = (Wl @ Wc).float().reshape(wl.shape[0], C_, 1, 1)
        return e.conv(d.st, self._pack(name + "@st", [(w2, one, zero)], C_), 1, 0, "none", res=t)

    # ------------------------------------------------------------------ neck
    def fpn(self, p: str, inputs: Sequence[TView], start_level: int = 0, num_outs: int = 5,
            add_extra_convs="on_output", relu_before_extra_convs: bool = False) -> List[TView]:
        """fpn.py:150-205, norm_cfg=None / act_cfg=None (conv + bias only)."""
        e = self.e
        if relu_before_extra_convs:
            raise NotImplementedError("FPN relu_before_extra_convs is not lowered")
        n_lat = len(inputs) - start_level
        conv = lambda name, x, stride, pad: e.conv(x, self._pack(name, [self._plain_part(name)], x.c), stride, pad, "none")
        inputs = list(inputs)
        mode = ("on_input" if add_extra_convs is True else add_extra_convs) if num_outs > n_lat else None
        # A/B switch for measurements: how much of the plug-in's linear tail rides in its per-window matrices
        #   "window" (default): all of it, gl_lateral;  "deferred": channel_conv composed with the lateral conv, per pixel;
        #   "stored": the plug-in output as a tensor, then the lateral conv
        tail = "stored" if os.environ.get("GLSDET_NO_LATERAL_FOLD") else os.environ.get("GLSDET_GL_TAIL", "window")
        if self.trace is not None:
            tail = "stored"          # (a trace holds the plug-in's output as the tensor the reference has there)
        conv = lambda name, x, stride, pad, _c=conv: self._rec(name, _c(name, x, stride, pad))
        for i, inp in enumerate(inputs):
            if isinstance(inp, GlPending):
                last = i == len(inputs) - 1
                if i < start_level:
                    inputs[i] = None                                         # a level the FPN does not read at all
                elif (mode == "on_input" and last) or (inp.p + ".channel_conv.weight") not in self.sd or tail == "stored":
                    inputs[i] = self._rec(inp.p, self.gl_fusion(inp.p, inp.x, inp.assoc))   # stored: the extra level reads C5 itself / BaseConv channel_cat
                elif tail == "deferred":
                    inputs[i] = self.gl_fusion(inp.p, inp.x, inp.assoc, defer=True)
        lat = []
        for i in range(n_lat):
            inp, name = inputs[i + start_level], "%s.lateral_convs.%d.conv" % (p, i)
            if isinstance(inp, GlPending):
                lat.append(self.gl_lateral(inp.p, inp.x, name, inp.assoc))
            elif isinstance(inp, GlDeferred):
                lat.append(self._lateral_of_deferred(name, inp))
            else:
                lat.append(conv(name, inp, 1, 0))
        for i in range(n_lat - 1, 0, -1):
            e.upsample_add(lat[i], lat[i - 1])
            self._rec("%s.topdown.%d" % (p, i - 1), lat[i - 1])
        outs = [conv("%s.fpn_convs.%d.conv" % (p, i), lat[i], 1, 1) for i in range(n_lat)]
        if num_outs > len(outs):
            if not add_extra_convs:
                for _ in range(num_outs - n_lat):
                    outs.append(e.pool2d(outs[-1], 1, 2, 0))
            else:
                mode = "on_input" if add_extra_convs is True else add_extra_convs
                src = {"on_input": inputs[-1], "on_lateral": lat[-1], "on_output": outs[-1]}[mode]
                outs.append(conv("%s.fpn_convs.%d.conv" % (p, n_lat), src, 2, 1))
                for i in range(n_lat + 1, num_outs):
                    outs.append(conv("%s.fpn_convs.%d.conv" % (p, i), outs[-1], 2, 1))
        return outs

    # ------------------------------------------------------------------ heads
    @staticmethod
    def _level_groups(feats: Sequence[TView], per_level: int = 1) -> List[List[int]]:
        """Levels whose maps are small share grouped launches (<= 4 convs per launch); a big level
        fills the chip alone."""
        big = [i for i, f in enumerate(feats) if f.h * f.w > 64 * 64 // 2]
        small = [i for i, f in enumerate(feats) if f.h * f.w <= 64 * 64 // 2]
        groups = [[i] for i in big]
        step = max(1, 4 // per_level)
        groups += [small[i:i + step] for i in range(0, len(small), step)]
        return groups

    def _conv_levels(self, xs: Sequence[TView], packs, pad: int, out_dtype=None, per_level: int = 1, gn_groups: int = 0):
        """One conv per entry (entries ordered level-major, `per_level` consecutive entries per level),
        grouped over the small levels.  gn_groups > 0: a GroupNorm of that many groups follows; the convs of the big
        levels then also write its partial sums (Engine.conv_gnstats) and (outs, stats) is returned, stats[i] None where
        the GroupNorm still has to sum the tensor itself."""
        L = len(xs) // per_level
        outs: List[Optional[TView]] = [None] * len(xs)
        stats: List[Optional[torch.Tensor]] = [None] * len(xs)
        for g in self._level_groups(xs[::per_level], per_level):
            idx = [l * per_level + j for l in g for j in range(per_level)]
            if gn_groups and len(g) == 1 and out_dtype is None:        # a big level: its convs run alone anyway
                for i in idx:
                    outs[i], stats[i] = self.e.conv_gnstats(xs[i], packs[i], pad, gn_groups)
                continue
            res = self.e.conv_group([xs[i] for i in idx], [packs[i] for i in idx], 1, pad, "none", out_dtype=out_dtype)
            for i, r in zip(idx, res):
                outs[i] = r
        return (outs, stats) if gn_groups else outs

    def _gn_params(self, key: str, names: Sequence[str]):
        ga = self._dev(key + ".gamma", torch.cat([self.sd[n + ".gn.weight"] for n in names]))
        be = self._dev(key + ".beta", torch.cat([self.sd[n + ".gn.bias"] for n in names]))
        return ga, be

    def towers(self, p: str, feats: Sequence[TView], stacked: int) -> Tuple[List[TView], List[TView]]:
        """cls / reg towers of gfl_head.py:128-152 for ALL levels (shared weights), emitted layer by
        layer: `stacked` x (3x3 conv, GN32, ReLU) each.  Layer 0 of both towers is one fused GEMM
        (Cout 512) + one 64-group GN; per layer the GroupNorms of all levels and both towers are
        one launch pair, the convs of the small levels share grouped launches."""
        e, L = self.e, len(feats)
        raw = lambda n: (self.sd[n + ".conv.weight"], torch.ones(self.sd[n + ".conv.weight"].shape[0]),
                         torch.zeros(self.sd[n + ".conv.weight"].shape[0]))
        names0 = ["%s.cls_convs.0" % p, "%s.reg_convs.0" % p]
        f = self.sd[names0[0] + ".conv.weight"].shape[0]
        pk0 = self._pack(p + ".tower0", [raw(n) for n in names0], feats[0].c)
        both, pre = self._conv_levels(feats, [pk0] * L, 1, gn_groups=64)
        ga, be = self._gn_params(p + ".tower0", names0)
        rec = lambda i, kind, ts: [self._rec("%s.%s_convs.%d.%s@%d" % (p, ("cls", "reg")[k % 2], i, kind, k // 2), t)
                                   for k, t in enumerate(ts)] if self.trace is not None else None
        cur = [t.channels(j * f, (j + 1) * f) for t in both for j in (0, 1)]      # level-major: cls_l, reg_l
        rec(0, "conv", cur)
        e.groupnorm_multi(both, 64, [ga] * L, [be] * L, GN_EPS, "relu", pre=pre)
        rec(0, "gn", cur)
        for i in range(1, stacked):
            names = ["%s.%s_convs.%d" % (p, which, i) for which in ("cls", "reg")]
            pks = [self._pack(n, [raw(n)], f) for n in names]
            gb = [self._gn_params(n, [n]) for n in names]
            cur, pre = self._conv_levels(cur, pks * L, 1, per_level=2, gn_groups=32)
            rec(i, "conv", cur)
            e.groupnorm_multi(cur, 32, [gb[j][0] for _ in range(L) for j in (0, 1)],
                              [gb[j][1] for _ in range(L) for j in (0, 1)], GN_EPS, "relu", pre=pre)
            rec(i, "gn", cur)
        return cur[0::2], cur[1::2]

    def _reg_preds(self, p: str, regs: Sequence[TView]) -> List[TView]:
        pks = []
        for l, r in enumerate(regs):
            scale = float(self.sd["%s.scales.%d.scale" % (p, l)])          # mmcv Scale folded into the weights
            pks.append(self._pack("%s.gfl_reg@%d" % (p, l), [self._plain_part(p + ".gfl_reg", scale)], r.c))
        return self._conv_levels(regs, pks, 1, out_dtype=F32)

    def gfl_head(self, p: str, feats: Sequence[TView], stacked: int = 4) -> Tuple[List[TView], List[TView]]:
        """gfl_head.py:179-203 -> fp32 views: cls logits [n,h,w,nc], reg logits [n,h,w,4*(reg_max+1)]."""
        cs, rs = self.towers(p, feats, stacked)
        pk = self._pack(p + ".gfl_cls", [self._plain_part(p + ".gfl_cls")], cs[0].c)
        return self._conv_levels(cs, [pk] * len(cs), 1, out_dtype=F32), self._reg_preds(p, rs)

    def mp_head(self, p: str, feats: Sequence[TView], proxies_list: Sequence[int], gamma: float = 10.0,
                stacked: int = 4) -> Tuple[List[TView], List[TView]]:
        """mp_head.py:123-154 (eval branch) + forward_proxy :105-121."""
        e = self.e
        prox = self.sd[p + ".proxies"].float()
        assert prox.shape[0] == sum(proxies_list), "proxies_list does not match the proxies parameter"
        centers = prox / prox.norm(dim=1, keepdim=True).clamp_min(1e-12)        # F.normalize(p=2, dim=1)
        cs, rs = self.towers(p, feats, stacked)
        reg = self._reg_preds(p, rs)
        pkf = self._pack(p + ".gfl_cls_conv", [self._plain_part(p + ".gfl_cls_conv")], cs[0].c)
        fs = self._conv_levels(cs, [pkf] * len(cs), 1)
        for l, t in enumerate(fs):
            self._rec("%s.gfl_cls_conv@%d" % (p, l), t)
        w = centers.reshape(centers.shape[0], centers.shape[1], 1, 1)
        pkp = self._pack(p + ".proxies", [(w, torch.ones(w.shape[0]), torch.zeros(w.shape[0]))], fs[0].c)
        dots = self._conv_levels(fs, [pkp] * len(fs), 0, out_dtype=F32)
        return [e.proxy_scores(f, d, list(proxies_list), gamma) for f, d in zip(fs, dots)], reg


class _Compiled:
    __slots__ = ("img", "plan", "cls", "reg", "nb", "eng", "post", "graph_stream", "scale", "img_hw")


class HipGflDetector:
    """ResNet-50 + FPN + GFLHead ('gfl') or MPHead ('mpdet'); mmdet state-dict names.
    cfg keys: start_level, num_outs, add_extra_convs, stacked_convs, strides, reg_max,
    proxies_list, gamma, depth, out_indices (defaults = the GFL r50-FPN / MPDet configs)."""

    DEFAULTS = dict(start_level=1, num_outs=5, add_extra_convs="on_output", relu_before_extra_convs=False,
                    stacked_convs=4, strides=(8, 16, 32, 64, 128), reg_max=16, depth=50, out_indices=(0, 1, 2, 3),
                    proxies_list=(2, 3, 2, 5, 4, 8, 8, 4, 3, 3), gamma=10.0,
                    gl_levels=None,          # backbone outputs that get x + Patch_Conv_NonLocal_new(x) (None: those whose
                    gl_assoc="auto")         # neck.gl_fusion.<i>.* keys exist in the state_dict); association of its products

    def __init__(self, kind: str, state_dict, dtype: str = "f16", device: str = "cuda:0", autotune: bool = False, **cfg):
        if kind not in ("gfl", "mpdet"):
            raise ValueError("kind must be 'gfl' or 'mpdet'")
        unknown = set(cfg) - set(self.DEFAULTS)
        if unknown:
            raise TypeError("unknown options %s" % sorted(unknown))
        self.kind, self.dtype, self.device, self.autotune = kind, dtype, device, autotune
        self.cfg = dict(self.DEFAULTS, **cfg)
        self.sd = {k: v.detach().cpu() for k, v in state_dict.items()}
        if kind == "gfl":
            self.num_classes = self.sd["bbox_head.gfl_cls.weight"].shape[0]
        else:
            self.num_classes = len(self.cfg["proxies_list"])
        self._compiled: Dict[Tuple, _Compiled] = {}
        self._cache_lock = threading.Lock()       # plan cache: looked up / LRU-touched / evicted by several lanes' threads

    def _emit(self, eng: Engine, img: torch.Tensor, trace: Optional[dict] = None):
        b, c = ResDetBuilder(eng, self.sd), self.cfg
        b.trace = trace
        stages = b.resnet("backbone", img, c["depth"], c["out_indices"])
        levels = c["gl_levels"]
        if levels is None:
            levels = [i for i in range(len(stages)) if "neck.gl_fusion.%d.feat_patchconv_lt_nonlocal.theta.weight" % i in self.sd]
        for i in levels:                       # GLFusionFPN: the plug-in sits on the FPN inputs (C3..C5), before the laterals
            stages[i] = GlPending("neck.gl_fusion.%d" % i, stages[i], c["gl_assoc"])
        feats = b.fpn("neck", stages, c["start_level"], c["num_outs"],
                      c["add_extra_convs"], c["relu_before_extra_convs"])
        if self.kind == "gfl":
            return b.gfl_head("bbox_head", feats, c["stacked_convs"])
        return b.mp_head("bbox_head", feats, c["proxies_list"], c["gamma"], c["stacked_convs"])

    def compile(self, n: int, H: int, W: int, post: Optional[dict] = None, use_graph: bool = False,
                instance: int = 0) -> _Compiled:
        """post: None or dict(score_thr, iou_thr, nms_pre=1000, max_per_img=100, max_cand, rescale=False)
        (the reference's test_cfg keys, base_dense_head.py:168,295-298)."""
        key = (n, H, W, tuple(sorted(post.items())) if post else None, use_graph, instance)
        with self._cache_lock:
            if key in self._compiled:
                self._compiled[key] = self._compiled.pop(key)      # most recently used last
                return self._compiled[key]
        eng = Engine(self.dtype, self.device, autotune=self.autotune)
        c = _Compiled()
        c.eng, c.post, c.nb, c.scale = eng, post, None, None
        c.img = torch.zeros(n, 3, H, W, dtype=torch.float32, device=eng.device)
        c.img_hw = torch.tensor([[H, W]] * n, dtype=torch.float32, device=eng.device)
        if post is not None and post.get("rescale"):
            c.scale = torch.ones(n, 4, dtype=torch.float32, device=eng.device)
        c.plan = eng.new_plan()
        with c.plan:
            c.cls, c.reg = self._emit(eng, c.img)
            if post is not None:
                L = len(c.cls)
                biggest = max(l.h * l.w for l in c.cls) * self.num_classes
                c.nb = eng.gfl_buffers(n, L, min(post.get("max_cand", 8192), biggest), post.get("nms_pre", 1000),
                                       post.get("max_per_img", 100))
                eng.gfl_detect(c.cls, c.reg, self.cfg["strides"][:L], self.num_classes, self.cfg["reg_max"], H, W,
                               post["score_thr"], post["iou_thr"], c.nb, img_hw=c.img_hw, scale_factors=c.scale)
                if post.get("exchange_cap"):
                    eng.pack_detections(c.nb, int(post["exchange_cap"]))
        eng.save_tune_cache()
        c.graph_stream = None
        if use_graph:
            # one capture at a time, and no device-wide synchronisation around it: another host thread (a lane of
            # the two-stage pipeline) may be capturing or launching -- a hipDeviceSynchronize issued while any
            # stream captures invalidates that capture
            with _lib.CAPTURE_LOCK:
                c.plan.run()                     # warm-up outside capture (lazy module load, attributes)
                torch.cuda.current_stream(eng.device).synchronize()
                c.graph_stream = torch.cuda.Stream(device=eng.device)
                with torch.cuda.stream(c.graph_stream):
                    c.plan.capture(c.graph_stream)
                c.graph_stream.synchronize()
        # (two threads asking for the same key at once both build; the second insert wins, the first plan is dropped: correct, rare)
        # LRU cap on resident plans (each holds its activations, an NMS workspace and, today, its own copy of the packed
        # weights): the UFPMP fine stage compiles one plan per padded mosaic shape and lane -- without a cap HBM grows with
        # every new shape over a data set.  GLSDET_MAX_PLANS (default 24) plans stay; evicted ones are rebuilt on demand.
        import os as _os
        cap = int(_os.environ.get("GLSDET_MAX_PLANS", "24"))
        with self._cache_lock:
            self._compiled[key] = c
            while len(self._compiled) > max(1, cap):
                self._compiled.pop(next(iter(self._compiled)))
        return c

    def run(self, c: _Compiled, img: Optional[torch.Tensor] = None, img_hw: Optional[torch.Tensor] = None,
            scale: Optional[torch.Tensor] = None):
        def feed():
            if img is not None:
                c.img.copy_(img, non_blocking=True)
            if img_hw is not None:
                c.img_hw.copy_(img_hw, non_blocking=True)
            if scale is not None:
                c.scale.copy_(scale, non_blocking=True)
        if c.plan.captured:
            st, cur = c.graph_stream, torch.cuda.current_stream()
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                feed()
                c.plan.launch(st)
            cur.wait_stream(st)
            return
        feed()
        c.plan.run(None)

    @staticmethod
    def run_async(c: _Compiled):
        assert c.plan.captured, "run_async needs compile(..., use_graph=True)"
        c.plan.launch(c.graph_stream)

    def forward_traced(self, img: torch.Tensor):
        """forward_raw emitted EAGERLY (no plan) with a snapshot of every stored tensor: (cls, reg, {store point: NCHW fp32})
        under the names of oracle/mpdet_oracle.py.  The traced graph is the unfused one where a fusion removes a tensor the
        reference has (stem + pool stay one launch: one store; the plug-in's output is stored, not folded into the lateral)."""
        eng = Engine(self.dtype, self.device)
        tr: Dict[str, torch.Tensor] = {}
        cls, reg = self._emit(eng, img.to(eng.device, torch.float32).contiguous(), tr)
        torch.cuda.synchronize(eng.device)
        bins = 4 * (self.cfg["reg_max"] + 1)
        return [l.to_nchw(self.num_classes) for l in cls], [l.to_nchw(bins) for l in reg], tr

    def forward_raw(self, img: torch.Tensor) -> Tuple[List[torch.Tensor], List[torch.Tensor]]:
        """Reference-shaped head outputs: (cls_scores, bbox_preds), per level NCHW fp32
        ([B,nc,H_l,W_l], [B,4*(reg_max+1),H_l,W_l]) -- gfl_head.py:154-177."""
        n, _, H, W = img.shape
        c = self.compile(n, H, W)
        self.run(c, img.to(c.img.device, torch.float32))
        bins = 4 * (self.cfg["reg_max"] + 1)
        return [l.to_nchw(self.num_classes) for l in c.cls], [l.to_nchw(bins) for l in c.reg]

    def detect(self, img: torch.Tensor, score_thr: float = 0.05, iou_thr: float = 0.6, nms_pre: int = 1000,
               max_per_img: int = 100, img_shapes=None, scale_factors=None, use_graph: bool = False, instance: int = 0):
        """-> list per image of (dets ndarray(k,5) x1,y1,x2,y2,score ; labels ndarray(k) int64).
        use_graph: replay a hipGraph captured for this (batch, H, W, thresholds) -- one plan per input shape
        stays resident (the fine stage of UFPMP-Det sees a few dozen padded mosaic shapes; at ~0.5 GB of
        activations each they all fit the HBM many times over)."""
        n, _, H, W = img.shape
        post = dict(score_thr=score_thr, iou_thr=iou_thr, nms_pre=nms_pre, max_per_img=max_per_img,
                    rescale=scale_factors is not None)
        c = self.compile(n, H, W, post, use_graph=use_graph, instance=instance)   # instance: private buffers per caller thread
        hw = None if img_shapes is None else torch.tensor([[s[0], s[1]] for s in img_shapes], dtype=torch.float32)
        sf = None if scale_factors is None else torch.tensor(np.asarray(scale_factors, np.float32).reshape(n, 4))
        self.run(c, img.to(c.img.device, torch.float32), None if hw is None else hw.to(c.img.device),
                 None if sf is None else sf.to(c.img.device))
        return self.collect(c)

    @staticmethod
    def collect(c: _Compiled):
        count = c.nb["count"].cpu().numpy()              # the one host sync of the path
        if int(c.nb["status"].item()) & 1:
            raise RuntimeError("GFL candidate capacity exceeded on a level (raise max_cand)")
        dets = c.nb["dets"].cpu().numpy()
        n = c.nb["n"]
        out = []
        for i in range(n):
            d = dets[i, : count[i]]
            out.append((d[:, :5].copy(), d[:, 6].astype(np.int64)))
        return out
