"""Graph builders: lower a GLSDet detector (given as a reference-named state_dict) to a
sequence of libglsdet_hip ops on NHWC views.

What the reference does with separate tensors and ``torch.cat`` / slicing is done here
with views into shared buffers, so no concat, quadrant slice or re-stitch ever moves data:

* every concat input is written by its producer straight into its channel slice;
* CSP ``conv1``/``conv2`` (same input, both 1x1) run as ONE GEMM with Cout = 2*hidden that
  lands directly in the concat buffer ``conv3`` reads; bottlenecks update their half in place;
* SPP pools 5/9/13 are three chained 5x5 pools (exact for stride-1 max pools);
* the four quadrant convs of a Patch_Conv write into one full-size buffer whose
  left/right/top/bottom halves ARE the reference's l/r/t/b stitches;
* the first convs of the cls and reg towers share their input and run as one GEMM; the three
  predictors of a level run as one 1x1 GEMM with block-structured weights that emits the
  reference's ``cat([reg, obj, cls])`` channel order (fp32 logits).

Reference topology: drone/models/base/{darknet,yolox}.py,
drone/models/block/non_local/{Identity_Conv,yolo_patch_nonlocal_plus}.py.
"""
from __future__ import annotations

import os

from typing import Dict, List, Optional, Sequence

import torch

from ._lib import F32
from .engine import Engine, TView, ceil_to, fold_bn

BN_EPS = 1e-3   # drone/models/base/baseConv.py:12


class NetBuilder:
    def __init__(self, eng: Engine, sd: Dict[str, torch.Tensor]):
        self.e = eng
        self.sd = {k: v.detach().cpu() for k, v in sd.items()}
        self._packed = {}
        self._override = {}     # BaseConv name -> (w, scale, bias) replacing the state_dict's (compose_1x1_input)
        self.trace = None       # tests: dict name -> NCHW fp32 snapshot of every stored tensor (eager emission only)
        self.composed = None    # tests: dict filled with name -> (w, scale, bias) of every host-composed conv; when set, a
        #                         trace keeps the composed forms (the tensors they remove are then simply not in the trace)

    def _rec(self, name: str, v: TView, c0: int = 0, c1: Optional[int] = None):
        """Per-layer trace for the parity tests (tests/test_f16_emulation.py): a snapshot of the tensor the
        op at `name` just stored.  Only meaningful when ops are emitted eagerly (outside a plan)."""
        if self.trace is not None:
            t = v.to_nchw()
            self.trace[name] = t[:, c0:(c1 if c1 is not None else t.shape[1])].cpu()

    # ------------------------------------------------------------------ weights
    def _bn_part(self, p: str):
        if p in self._override:            # a BaseConv whose weights were composed with the linear conv in front of it
            return self._override[p]
        s, b = fold_bn(self.sd[p + ".bn.weight"], self.sd[p + ".bn.bias"],
                       self.sd[p + ".bn.running_mean"], self.sd[p + ".bn.running_var"], BN_EPS)
        return self.sd[p + ".conv.weight"], s, b

    def _plain_part(self, p: str):
        w = self.sd[p + ".weight"]
        b = self.sd.get(p + ".bias")
        if b is None:
            b = torch.zeros(w.shape[0])
        return w, torch.ones(w.shape[0]), b

    def _pack(self, key, parts, cin_pad):
        k = (key, cin_pad)
        if k not in self._packed:
            self._packed[k] = self.e.pack_conv(parts, cin_pad)
        return self._packed[k]

    def has(self, key: str) -> bool:
        return key in self.sd

    def is_depthwise(self, p: str) -> bool:
        return (p + ".dconv.conv.weight") in self.sd

    # ------------------------------------------------------------------ conv flavours
    def cba(self, prefixes, x: TView, stride: int = 1, act: str = "silu", out: Optional[TView] = None,
            res: Optional[TView] = None) -> TView:
        """BaseConv(s): conv (no bias) + BN + act; several prefixes = fused along Cout."""
        if isinstance(prefixes, str):
            prefixes = [prefixes]
        if any(self.is_depthwise(p) for p in prefixes):
            return self._dw_separable(prefixes, x, stride, act, out, res)
        parts = [self._bn_part(p) for p in prefixes]
        k = parts[0][0].shape[-1]
        pk = self._pack("+".join(prefixes), parts, x.c)
        y = self.e.conv(x, pk, stride, (k - 1) // 2, act, out=out, res=res)
        c0 = 0
        for p, pt in zip(prefixes, parts):
            self._rec(p, y, c0, c0 + pt[0].shape[0])
            c0 += pt[0].shape[0]
        return y

    def cba_group(self, prefixes: Sequence[str], xs: Sequence[TView], stride: int = 1, act: str = "silu",
                  outs: Optional[Sequence[Optional[TView]]] = None) -> List[TView]:
        """Independent BaseConvs of one shape class (quadrant convs, the cls / reg tower convs of a
        level): one grouped launch where that is faster (Engine.conv_group), same results."""
        outs = list(outs) if outs is not None else [None] * len(xs)
        if any(self.is_depthwise(p) for p in prefixes):
            return [self.cba(p, x, stride, act, out=o) for p, x, o in zip(prefixes, xs, outs)]
        parts = [self._bn_part(p) for p in prefixes]
        k = parts[0][0].shape[-1]
        packs = [self._pack(p, [pt], x.c) for p, pt, x in zip(prefixes, parts, xs)]
        ys = self.e.conv_group(xs, packs, stride, (k - 1) // 2, act, outs=outs)
        for p, pt, y in zip(prefixes, parts, ys):
            self._rec(p, y, 0, pt[0].shape[0])
        return ys

    def _dw_separable(self, prefixes, x: TView, stride, act, out, res) -> TView:
        """DWConv (baseConv.py:22-30): depthwise kxk (+BN+act) then pointwise 1x1 (+BN+act).
        A list of prefixes (outputs concatenated along C) is lowered one by one."""
        couts = [self.sd[p + ".pconv.conv.weight"].shape[0] for p in prefixes]
        if out is None:
            k0 = self.sd[prefixes[0] + ".dconv.conv.weight"].shape[-1]
            ho = (x.h + 2 * ((k0 - 1) // 2) - k0) // stride + 1
            wo = (x.w + 2 * ((k0 - 1) // 2) - k0) // stride + 1
            out = self.e.tensor(x.n, ho, wo, sum(couts))
        c0 = 0
        for p, co in zip(prefixes, couts):
            w, s, b = self._bn_part(p + ".dconv")
            k = w.shape[-1]
            key = (p + ".dconv", x.c)
            if key not in self._packed:
                self._packed[key] = self.e.pack_dw(w, s, b, x.c)
            t = self.e.dwconv(x, self._packed[key], stride, (k - 1) // 2, act)
            self._rec(p + ".dconv", t, 0, w.shape[0])
            dst = out if len(prefixes) == 1 else out.channels(c0, c0 + co)
            pk = self._pack(p + ".pconv", [self._bn_part(p + ".pconv")], t.c)
            self.e.conv(t, pk, 1, 0, act, out=dst, res=res)
            self._rec(p + ".pconv", dst, 0, co)
            c0 += co
        return out

    def plain(self, p: str, x: TView, pad: int = 0, out: Optional[TView] = None, act: str = "none",
              res: Optional[TView] = None) -> TView:
        """nn.Conv2d with bias, no norm; activation / residual add only where the caller fuses one."""
        pk = self._pack(p, [self._plain_part(p)], x.c)
        y = self.e.conv(x, pk, 1, pad, act, out=out, res=res)
        self._rec(p, y, 0, pk[3])
        return y

    def conv_out_channels(self, p: str) -> int:
        if self.is_depthwise(p):
            return self.sd[p + ".pconv.conv.weight"].shape[0]
        return self.sd[p + ".conv.weight"].shape[0]

    # ------------------------------------------------------------------ blocks
    def _cba_chain(self, prefixes, x: TView, nxt: str, cin2: int, out: TView, res: Optional[TView] = None) -> Optional[TView]:
        """BaseConv(s) `prefixes` (as cba, written to `out`) with the 1x1 BaseConv `nxt` chained onto channels [0, cin2) of
        the result in the same launch (Engine.conv_chain).  -> nxt's output, or None when the fused form does not apply
        (nothing was emitted then)."""
        if isinstance(prefixes, str):
            prefixes = [prefixes]
        if os.environ.get("GLSDET_NO_CHAIN") or any(self.is_depthwise(q) for q in list(prefixes) + [nxt]):
            return None
        parts = [self._bn_part(q) for q in prefixes]
        k = parts[0][0].shape[-1]
        pk = self._pack("+".join(prefixes), parts, x.c)
        w2 = self.sd[nxt + ".conv.weight"]
        if w2.shape[-1] != 1 or w2.shape[1] != cin2:
            return None
        pk2 = self._pack(nxt, [self._bn_part(nxt)], cin2)
        t = self.e.tensor(x.n, out.h, out.w, w2.shape[0])
        if not self.e.conv_chain(x, pk, 1, (k - 1) // 2, "silu", out, res, pk2, "silu", 0, cin2, t):
            return None
        c0 = 0
        for q, pt in zip(prefixes, parts):
            self._rec(q, out, c0, c0 + pt[0].shape[0])
            c0 += pt[0].shape[0]
        self._rec(nxt, t, 0, w2.shape[0])
        return t

    def _csp_entry(self, prefixes, x: Optional[TView], dst: TView, down):
        """conv1 | conv2 of a CSPLayer (in the order `prefixes`) -> dst.  down = (prefix, input) of the 3x3 stride-2 BaseConv
        that produces the layer's input x and is read by nothing else (darknet.py:174-195): where the chained kernel takes
        the pair (and beats the two tuned launches), conv1 | conv2 ride in ITS launch and its output is never stored
        (Engine.conv_chain(skip_y=True)); otherwise it is emitted first.  -> None (dst is written either way)."""
        if down is not None:
            dp, dx = down
            cin2 = self.conv_out_channels(dp)
            hid2 = sum(self.conv_out_channels(q) for q in prefixes)
            ok = False
            if not os.environ.get("GLSDET_NO_DOWN_CHAIN") and not os.environ.get("GLSDET_NO_CHAIN") and hid2 <= 128 and \
                    not any(self.is_depthwise(q) for q in list(prefixes) + [dp]):
                pk = self._pack(dp, [self._bn_part(dp)], dx.c)
                pk2 = self._pack("+".join(prefixes), [self._bn_part(q) for q in prefixes], ceil_to(cin2, 8))
                ho, wo = (dx.h + 2 - 3) // 2 + 1, (dx.w + 2 - 3) // 2 + 1
                scratch = self.e.tensor(dx.n, ho, wo, cin2)
                ok = self.e.conv_chain(dx, pk, 2, 1, "silu", scratch, None, pk2, "silu", 0, scratch.c, dst, skip_y=True)
            if ok:
                return
            x = self.cba(dp, dx, 2)
        self.cba(list(prefixes), x, out=dst)

    def csp(self, p: str, x: Optional[TView], shortcut: bool, out: Optional[TView] = None, down=None) -> TView:
        """CSPLayer (darknet.py:66-112).  down: see _csp_entry (x is then None: the layer's input exists only inside the
        producing launch).  Where the fused kernel applies and pays, a Bottleneck (1x1 -> 3x3 [+ x],
        darknet.py:61-64) is ONE launch that recomputes the 1x1 on the 3x3's halo (glsdet_bottleneck): the hidden tensor
        never reaches memory.  That form cannot run in place, so the main branch ping-pongs between two channel slots
        P | Q of one buffer [P | short | Q]; conv1|conv2 write [main | short] into [P | short] for an even number of
        Bottlenecks and, with the two weight blocks swapped, [short | main] into [short | Q] for an odd one, so that the
        last Bottleneck always lands in P and conv3 reads [P | short] as the reference's torch.cat.
        Otherwise every 1x1 `m.i.conv1` rides in the launch of the conv that produces its input (conv1|conv2 for i = 0,
        Bottleneck i-1's 3x3 otherwise) where the chained kernel applies (glsdet_conv2d_chain)."""
        hid = self.conv_out_channels(p + ".conv1")
        n = 0
        while self.has("%s.m.%d.conv1.conv.weight" % (p, n)):
            n += 1
        # (tracing keeps the unfused form: a trace holds EVERY stored tensor, and the fused form equals it bit for bit)
        if n and hid in (32, 64, 128) and not os.environ.get("GLSDET_NO_BNECK_FUSION") and self.trace is None and \
                not self.is_depthwise("%s.m.0.conv2" % p) and self.sd["%s.m.0.conv2.conv.weight" % p].shape[-1] == 3:
            if x is None:
                dn, dh, dw = down[1].n, (down[1].h + 2 - 3) // 2 + 1, (down[1].w + 2 - 3) // 2 + 1
            else:
                dn, dh, dw = x.n, x.h, x.w
            buf = self.e.tensor(dn, dh, dw, 3 * hid)
            P, Q = buf.channels(0, hid), buf.channels(2 * hid, 3 * hid)
            scratch = self.e.tensor(dn, dh, dw, hid)
            cur, oth = (P, Q) if n % 2 == 0 else (Q, P)
            probe = lambda i, src, dst: self.e.bottleneck(src, self._pack("%s.m.%d.conv1" % (p, i), [self._bn_part("%s.m.%d.conv1" % (p, i))], hid),
                                                          "silu", self._pack("%s.m.%d.conv2" % (p, i), [self._bn_part("%s.m.%d.conv2" % (p, i))], hid),
                                                          "silu", dst, src if shortcut else None, scratch)
            if n % 2 == 0:
                self._csp_entry([p + ".conv1", p + ".conv2"], x, buf.channels(0, 2 * hid), down)          # [main | short] -> [P | short]
            else:
                self._csp_entry([p + ".conv2", p + ".conv1"], x, buf.channels(hid, 3 * hid), down)        # [short | main] -> [short | Q]
            for i in range(n):
                if not probe(i, cur, oth):
                    t = self.cba("%s.m.%d.conv1" % (p, i), cur, out=scratch)
                    self.cba("%s.m.%d.conv2" % (p, i), t, out=oth, res=cur if shortcut else None)
                else:
                    self._rec("%s.m.%d.conv2" % (p, i), oth, 0, hid)
                cur, oth = oth, cur
            return self.cba(p + ".conv3", buf.channels(0, 2 * hid), out=out)
        if x is None and down is not None and not (self.trace is None and 2 * hid <= 128):
            x, down = self.cba(down[0], down[1], 2), None               # (wider layers / tracing: the producer is emitted on its own)
        if x is None:
            cat = self.e.tensor(down[1].n, (down[1].h + 2 - 3) // 2 + 1, (down[1].w + 2 - 3) // 2 + 1, 2 * hid)
            self._csp_entry([p + ".conv1", p + ".conv2"], None, cat, down)
            t = None
        else:
            cat = self.e.tensor(x.n, x.h, x.w, 2 * hid)
            t = self._cba_chain([p + ".conv1", p + ".conv2"], x, "%s.m.0.conv1" % p, hid, cat) if n else None
            if t is None:
                self.cba([p + ".conv1", p + ".conv2"], x, out=cat)          # [main | short]
        a = cat.channels(0, hid)
        for i in range(n):
            if t is None:
                t = self.cba("%s.m.%d.conv1" % (p, i), a)
            t_next = None
            if i + 1 < n:
                t_next = self._cba_chain("%s.m.%d.conv2" % (p, i), t, "%s.m.%d.conv1" % (p, i + 1), hid, a,
                                         res=a if shortcut else None)
            if t_next is None:
                self.cba("%s.m.%d.conv2" % (p, i), t, out=a, res=a if shortcut else None)   # in place
            t = t_next
        return self.cba(p + ".conv3", cat, out=out)

    def spp(self, p: str, x: TView, out: Optional[TView] = None) -> TView:
        hid = self.conv_out_channels(p + ".conv1")
        cat = self.e.tensor(x.n, x.h, x.w, 4 * hid)
        self.cba(p + ".conv1", x, out=cat.channels(0, hid))
        if os.environ.get("GLSDET_NO_SPP_FUSION"):
            for i in range(3):      # 5 -> 9 -> 13 by chaining 5x5 pools
                self.e.maxpool(cat.channels(i * hid, (i + 1) * hid), 5, out=cat.channels((i + 1) * hid, (i + 2) * hid))
        else:                       # the same three pools in one launch (glsdet_spp_pools)
            self.e.spp_pools(cat.channels(0, hid), cat.channels(hid, 2 * hid), cat.channels(2 * hid, 3 * hid), cat.channels(3 * hid, 4 * hid))
        return self.cba(p + ".conv2", cat, out=out)

    def darknet(self, p: str, img: torch.Tensor, homes: Dict[str, Optional[TView]]) -> Dict[str, TView]:
        """CSPDarknet.  homes[name] = view the named output must be written to (or None)."""
        f = {}
        q = p + ".stem.conv"
        w0 = self.sd.get(q + ".conv.weight")
        d2 = p + ".dark2.0"
        wd = self.sd.get(d2 + ".conv.weight")
        fused_down = False
        if w0 is not None and img.shape[1] == 3 and tuple(w0.shape[1:]) == (12, 3, 3) and w0.shape[0] <= 64 and \
                not os.environ.get("GLSDET_NO_STEM_FUSION"):
            if w0.shape[0] == 32 and wd is not None and tuple(wd.shape[1:]) == (32, 3, 3) and wd.shape[0] <= 64 and self.trace is None \
                    and img.shape[2] % 4 == 0 and img.shape[3] % 4 == 0 and os.environ.get("GLSDET_STEM2_FUSION"):
                # ... and dark2.0 (3x3 stride 2) too: the stem's output, the largest tensor of the net, never exists either
                # (glsdet_focus_conv_down; equals the two-launch form bit for bit).  OFF unless GLSDET_STEM2_FUSION=1: measured
                # 198 us against 60 + 67 for the two launches (tools/probe/stem2_knock.py)
                x = self.e.focus_conv_down(img, self._pack(q, [self._bn_part(q)], 16), "silu", self._pack(d2, [self._bn_part(d2)], 32), "silu")
                fused_down = True
            else:
                # Focus + stem conv in one launch: the packed tensor is never written (glsdet_focus_conv)
                x = self.e.focus_conv(img, self._pack(q, [self._bn_part(q)], 16), "silu")
                self._rec(q, x, 0, w0.shape[0])
        else:
            x = self.cba(q, self.e.focus_pack(img))
        att = lambda i: self.has("%s.lsk%d.proj_1.weight" % (p, i))     # new/darknet_att.py:161-201
        for i, name in enumerate(("dark2", "dark3", "dark4")):
            down = None
            if not (fused_down and name == "dark2"):
                d0 = "%s.%s.0" % (p, name)
                # the stride-2 conv feeds the CSPLayer alone: where the layer's conv1 | conv2 fit the chained kernel (<= 128
                # output channels: dark2 / dark3 of YOLOX-s) they ride in its launch and its output never reaches memory
                if self.trace is None and not self.is_depthwise(d0) and 2 * self.conv_out_channels("%s.%s.1.conv1" % (p, name)) <= 128 \
                        and self.sd[d0 + ".conv.weight"].shape[-1] == 3:
                    down = (d0, x)
                    x = None
                else:
                    x = self.cba(d0, x, 2)
            if att(i + 2):
                x = self.csp("%s.%s.1" % (p, name), x, True, down=down)
                x = self.attention("%s.lsk%d" % (p, i + 2), x, out=homes.get(name))
            else:
                x = self.csp("%s.%s.1" % (p, name), x, True, out=homes.get(name), down=down)
            f[name] = x
        x = self.cba(p + ".dark5.0", x, 2)
        x = self.spp(p + ".dark5.1", x)
        if att(5):
            x = self.csp(p + ".dark5.2", x, False)
            x = self.attention(p + ".lsk5", x, out=homes.get("dark5"))
        else:
            x = self.csp(p + ".dark5.2", x, False, out=homes.get("dark5"))
        f["dark5"] = x
        return f

    def nonlocal_block(self, p: str, x: TView) -> TView:
        """In place on x.  theta|phi|g as one 1x1 GEMM, then the re-associated bilinear form."""
        ci = self.sd[p + ".theta.weight"].shape[0]
        key = p + ".tpg"
        parts = [self._plain_part(p + ".theta"), self._plain_part(p + ".phi"), self._plain_part(p + ".g")]
        tpg = self.e.conv(x, self._pack(key, parts, x.c), 1, 0, "none")
        for j, nm in enumerate(("theta", "phi", "g")):
            self._rec("%s.%s" % (p, nm), tpg, j * ci, (j + 1) * ci)
        if key + ".out" not in self._packed:
            w = self.sd[p + ".conv_out.weight"].float().reshape(-1, ci)
            self._packed[key + ".out"] = (self.e.upload(w), self.e.upload(self.sd[p + ".conv_out.bias"].float()))
        wout, bout = self._packed[key + ".out"]
        assert wout.shape[0] == x.c, "Non_local_Block conv_out must map back to the input channels"
        y = self.e.nonlocal_(x, tpg, ci, wout, bout, out=x)
        self._rec(p + ".conv_out", y)
        return y

    def nonlocal_blocks(self, ps: Sequence[str], xs: Sequence[TView]) -> List[TView]:
        """Several independent non-local blocks (the four quadrants), each in place on its x: the
        theta|phi|g projections go out as one grouped launch."""
        ci = self.sd[ps[0] + ".theta.weight"].shape[0]
        packs = []
        for p, x in zip(ps, xs):
            parts = [self._plain_part(p + ".theta"), self._plain_part(p + ".phi"), self._plain_part(p + ".g")]
            packs.append(self._pack(p + ".tpg", parts, x.c))
        tpgs = self.e.conv_group(xs, packs, 1, 0, "none")
        for p, tpg in zip(ps, tpgs):
            for j, nm in enumerate(("theta", "phi", "g")):
                self._rec("%s.%s" % (p, nm), tpg, j * ci, (j + 1) * ci)
        wouts, bouts = [], []
        for p, x in zip(ps, xs):
            key = p + ".tpg"
            if key + ".out" not in self._packed:
                w = self.sd[p + ".conv_out.weight"].float().reshape(-1, ci)
                self._packed[key + ".out"] = (self.e.upload(w), self.e.upload(self.sd[p + ".conv_out.bias"].float()))
            wout, bout = self._packed[key + ".out"]
            assert wout.shape[0] == x.c, "Non_local_Block conv_out must map back to the input channels"
            wouts.append(wout)
            bouts.append(bout)
        if len({(x.n, x.c) for x in xs}) == 1 and len(xs) <= 4 and not os.environ.get("GLSDET_NO_GROUP"):
            ys = self.e.nonlocal_multi(xs, tpgs, ci, wouts, bouts)
        else:
            ys = [self.e.nonlocal_(x, tpg, ci, w, b, out=x) for x, tpg, w, b in zip(xs, tpgs, wouts, bouts)]
        for p, y in zip(ps, ys):
            self._rec(p + ".conv_out", y)
        return ys

    def compose_1x1_input(self, consumer: str, c0: int, c1: int, lin: str):
        """The 1x1 BaseConv `consumer` reads channels [c0, c1) of its (concatenated) input from the linear 1x1 conv `lin`
        (weight + bias, nothing in between): replace them by lin's OWN input channels, W' = [W[:, :c0] | W[:, c0:c1] Wlin |
        W[:, c1:]] and the bias through the BN fold -- lin is then never launched (float64 on the host, exact)."""
        w, s, b = self._bn_part(consumer)
        wl = self.sd[lin + ".weight"].double().reshape(self.sd[lin + ".weight"].shape[0], -1)
        bl = self.sd[lin + ".bias"].double()
        W = w.double().reshape(w.shape[0], -1)
        assert W.shape[1] >= c1 and c1 - c0 == wl.shape[0]
        Wn = torch.cat([W[:, :c0], W[:, c0:c1] @ wl, W[:, c1:]], 1)
        bn = b.double() + s.double() * (W[:, c0:c1] @ bl)
        self._override[consumer] = (Wn.float().reshape(w.shape[0], -1, 1, 1), s, bn.float())
        if self.composed is not None:
            self.composed[consumer] = self._override[consumer]

    def patch_conv(self, p: str, x: TView, stride: int, with_nonlocal: bool, out: Optional[TView] = None,
                   z_out: Optional[TView] = None) -> TView:
        """Patch_Conv / Patch_Conv_NonLocal (Identity_Conv.py:267-387).  z_out: write the [lr | tb] tensor there and stop
        before the linear channel_conv (the caller composed it with its readers: compose_1x1_input)."""
        mid = self.conv_out_channels(p + ".feat_patchconv_lt")
        hh, hw = x.h // 2, x.w // 2               # int(H/2): floor split
        osz = lambda v: (v + 2 - 3) // stride + 1
        ht, hb, wl, wr = osz(hh), osz(x.h - hh), osz(hw), osz(x.w - hw)
        Q = self.e.tensor(x.n, ht + hb, wl + wr, mid)
        quads = {"lt": (x.window(0, hh, 0, hw), Q.window(0, ht, 0, wl)),
                 "lb": (x.window(hh, x.h, 0, hw), Q.window(ht, ht + hb, 0, wl)),
                 "rt": (x.window(0, hh, hw, x.w), Q.window(0, ht, wl, wl + wr)),
                 "rb": (x.window(hh, x.h, hw, x.w), Q.window(ht, ht + hb, wl, wl + wr))}
        names = list(quads)
        # the four quadrant convs are independent and of one shape class: one grouped launch
        self.cba_group(["%s.feat_patchconv_%s" % (p, nm) for nm in names], [quads[nm][0] for nm in names], stride,
                       outs=[quads[nm][1] for nm in names])
        if with_nonlocal:
            self.nonlocal_blocks(["%s.feat_patchconv_%s_nonlocal" % (p, nm) for nm in names],
                                 [quads[nm][1] for nm in names])
        H, W = ht + hb, wl + wr
        Z = self.e.tensor(x.n, H, W, 2 * mid) if z_out is None else z_out
        lr, tb = Z.channels(0, mid), Z.channels(mid, 2 * mid)
        self.cba_group([p + ".feat_patchconv_l", p + ".feat_patchconv_r"],
                       [Q.window(0, H, 0, wl), Q.window(0, H, wl, W)],
                       outs=[lr.window(0, H, 0, wl), lr.window(0, H, wl, W)])
        self.cba_group([p + ".feat_patchconv_t", p + ".feat_patchconv_b"],
                       [Q.window(0, ht, 0, W), Q.window(ht, H, 0, W)],
                       outs=[tb.window(0, ht, 0, W), tb.window(ht, H, 0, W)])
        if z_out is not None:
            return Z
        if self.has(p + ".channel_conv.weight"):
            return self.plain(p + ".channel_conv", Z, out=out)
        return self.cba(p + ".channel_conv", Z, out=out)

    def patch_conv_nonlocal_44(self, p: str, x: TView, out: Optional[TView] = None) -> TView:
        """Patch_Conv_NonLocal_44 (new/Non_local_family.py:359-421): a stride-2 Patch_Conv_NonLocal per quadrant, each
        writing its window of ONE buffer whose left / right / top / bottom halves are the reference's l / r / t / b
        stitches; a 1x1 BaseConv per half; channel_conv over [lr | tb]."""
        inner = lambda v: (v // 2 + 2 - 3) // 2 + 1 + ((v - v // 2) + 2 - 3) // 2 + 1      # patch_conv(stride 2) output extent
        hh, hw = x.h // 2, x.w // 2
        ht, hb, wl, wr = inner(hh), inner(x.h - hh), inner(hw), inner(x.w - hw)
        cq = self.sd[p + ".feat_patchconv_l.conv.weight"].shape[1]
        m = self.conv_out_channels(p + ".feat_patchconv_l")
        Q = self.e.tensor(x.n, ht + hb, wl + wr, cq)
        for nm, xin, dst in (("lt", x.window(0, hh, 0, hw), Q.window(0, ht, 0, wl)),
                             ("lb", x.window(hh, x.h, 0, hw), Q.window(ht, ht + hb, 0, wl)),
                             ("rt", x.window(0, hh, hw, x.w), Q.window(0, ht, wl, wl + wr)),
                             ("rb", x.window(hh, x.h, hw, x.w), Q.window(ht, ht + hb, wl, wl + wr))):
            self.patch_conv("%s.patchconv_%s_nonlocal" % (p, nm), xin, 2, True, out=dst)
        H, W = ht + hb, wl + wr
        Z = self.e.tensor(x.n, H, W, 2 * m)
        lr, tb = Z.channels(0, m), Z.channels(m, 2 * m)
        self.cba_group([p + ".feat_patchconv_l", p + ".feat_patchconv_r"], [Q.window(0, H, 0, wl), Q.window(0, H, wl, W)],
                       outs=[lr.window(0, H, 0, wl), lr.window(0, H, wl, W)])
        self.cba_group([p + ".feat_patchconv_t", p + ".feat_patchconv_b"], [Q.window(0, ht, 0, W), Q.window(ht, H, 0, W)],
                       outs=[tb.window(0, ht, 0, W), tb.window(ht, H, 0, W)])
        if self.has(p + ".channel_conv.weight"):
            return self.plain(p + ".channel_conv", Z, out=out)
        return self.cba(p + ".channel_conv", Z, out=out)

    def patch_conv_nonlocal_new(self, p: str, x: TView, out: Optional[TView] = None) -> TView:
        """Patch_Conv_NonLocal_new (new/Non_local_family.py:208-252).  The four quadrant
        non-local blocks run IN PLACE on windows of x (x is consumed), so the re-stitch is free."""
        hh, hw = x.h // 2, x.w // 2
        wins = (("lt", x.window(0, hh, 0, hw)), ("lb", x.window(hh, x.h, 0, hw)),
                ("rt", x.window(0, hh, hw, x.w)), ("rb", x.window(hh, x.h, hw, x.w)))
        self.nonlocal_blocks(["%s.feat_patchconv_%s_nonlocal" % (p, nm) for nm, _ in wins], [w for _, w in wins])
        if self.has(p + ".channel_conv.weight"):
            return self.plain(p + ".channel_conv", x, out=out)
        return self.cba(p + ".channel_conv", x, out=out)

    def patch_conv_nonlocal_adapt_new(self, p: str, x: TView, out: Optional[TView] = None) -> TView:
        """Patch_Conv_NonLocal_adapt_new (new/Non_local_family.py:272-357).  The quadrant split is computed ON the device
        from the thresholded attention map and stays there (Engine.attn_split): the four non-local blocks read their
        windows from it, the top / bottom 3x3 convs run on row-masked copies (the zeros are the padding the reference's
        sliced tensors see at the split) and a row select re-joins them -- no host synchronisation, where the reference
        makes one per column and row."""
        e = self.e
        att = self.spatial_attention(p + ".attention_map", x)
        split = e.attn_split(att)
        self.last_split = split
        names = ["%s.feat_patchconv_%s_nonlocal" % (p, q) for q in ("lt", "lb", "rt", "rb")]
        ci = self.sd[names[0] + ".theta.weight"].shape[0]
        packs, wouts, bouts = [], [], []
        for q in names:
            parts = [self._plain_part(q + ".theta"), self._plain_part(q + ".phi"), self._plain_part(q + ".g")]
            packs.append(self._pack(q + ".tpg", parts, x.c))
            key = q + ".tpg.out"
            if key not in self._packed:
                self._packed[key] = (e.upload(self.sd[q + ".conv_out.weight"].float().reshape(-1, ci)),
                                     e.upload(self.sd[q + ".conv_out.bias"].float()))
            wouts.append(self._packed[key][0])
            bouts.append(self._packed[key][1])
        tpgs = e.conv_group([x] * 4, packs, 1, 0, "none")       # every quadrant's projections over the full map
        S = e.nonlocal_split(x, tpgs, ci, wouts, bouts, e.tensor(x.n, x.h, x.w, x.c), split)
        T = self.cba(p + ".feat_patchconv_t", e.rowsplit(S, None, split, 0))
        B = self.cba(p + ".feat_patchconv_b", e.rowsplit(S, None, split, 1))
        Fm = e.rowsplit(T, B, split, 2)
        if self.has(p + ".channel_conv.weight"):
            y = self.plain(p + ".channel_conv", Fm)
        else:
            y = self.cba(p + ".channel_conv", Fm)
        return e.scale_by_map(y, att, out)

    def patch_conv_nonlocal_adapt(self, p: str, x: TView, out: Optional[TView] = None) -> TView:
        """Patch_Conv_NonLocal_adapt (new/Non_local_family.py:112-206): as the _new form, with a stride-2 3x3 BaseConv per
        quadrant in front of its non-local block and no gating.  The split indices are even (get_centroid rounds them), so a
        quadrant of the input maps onto the region [split / 2) of the stride-2 map: each quadrant conv runs over the whole
        map on a copy that is zero outside its quadrant (its padding at the split) and its region is merged into one buffer;
        everything downstream works on the halved split in device memory."""
        e = self.e
        if x.h % 2 or x.w % 2:
            raise ValueError("Patch_Conv_NonLocal_adapt needs even H and W (the reference's own torch.cat fails otherwise)")
        att = self.spatial_attention(p + ".attention_map", x)
        split = e.attn_split(att)
        self.last_split = split
        quads = ("lt", "lb", "rt", "rb")
        masked = [e.rowsplit(x, None, split, 3, quadrant=i) for i in range(4)]
        convs = self.cba_group(["%s.feat_patchconv_%s" % (p, q) for q in quads], masked, 2)
        Q = e.tensor(convs[0].n, convs[0].h, convs[0].w, convs[0].c)
        for i in range(4):
            e.rowsplit(convs[i], None, split, 4, out=Q, quadrant=i, shift=1)
        names = ["%s.feat_patchconv_%s_nonlocal" % (p, q) for q in quads]
        ci = self.sd[names[0] + ".theta.weight"].shape[0]
        packs, wouts, bouts = [], [], []
        for q in names:
            parts = [self._plain_part(q + ".theta"), self._plain_part(q + ".phi"), self._plain_part(q + ".g")]
            packs.append(self._pack(q + ".tpg", parts, Q.c))
            key = q + ".tpg.out"
            if key not in self._packed:
                self._packed[key] = (e.upload(self.sd[q + ".conv_out.weight"].float().reshape(-1, ci)),
                                     e.upload(self.sd[q + ".conv_out.bias"].float()))
            wouts.append(self._packed[key][0])
            bouts.append(self._packed[key][1])
        tpgs = e.conv_group([Q] * 4, packs, 1, 0, "none")
        S = e.nonlocal_split(Q, tpgs, ci, wouts, bouts, e.tensor(Q.n, Q.h, Q.w, Q.c), split, shift=1)
        T = self.cba(p + ".feat_patchconv_t", e.rowsplit(S, None, split, 0, shift=1))
        B = self.cba(p + ".feat_patchconv_b", e.rowsplit(S, None, split, 1, shift=1))
        Fm = e.rowsplit(T, B, split, 2, shift=1)
        if self.has(p + ".channel_conv.weight"):
            return self.plain(p + ".channel_conv", Fm, out=out)
        return self.cba(p + ".channel_conv", Fm, out=out)

    def lsk_block(self, p: str, x: TView, out: Optional[TView] = None) -> TView:
        """LSKblock (drone/models/lsk/LSK.py:27-51): depthwise 5x5, depthwise 7x7 with dilation 3 (glsdet_dwconv2d_dilated),
        conv1 / conv2 (1x1 to dim / 2, one grouped launch) into the two halves of ONE buffer = the reference's torch.cat,
        channel mean / max of it, 7x7 conv 2 -> 2 + sigmoid (its input channels swapped at pack time: the reference
        concatenates [mean, max], glsdet_channel_maxmean writes [max, mean]), glsdet_gate for
        attn1 * sig0 + attn2 * sig1, 1x1 to dim, glsdet_gate for x * attn."""
        e = self.e
        dim = self.sd[p + ".conv0.weight"].shape[0]
        half = self.sd[p + ".conv1.weight"].shape[0]

        def dw(q):
            key = (q, "dw")
            if key not in self._packed:
                w = self.sd[q + ".weight"]
                self._packed[key] = e.pack_dw(w, torch.ones(w.shape[0]), self.sd[q + ".bias"], x.c)
            return self._packed[key]
        a1 = e.dwconv(x, dw(p + ".conv0"), 1, 2, "none")
        self._rec(p + ".conv0", a1, 0, dim)
        a2 = e.dwconv(a1, dw(p + ".conv_spatial"), 1, 9, "none", dilation=3)
        self._rec(p + ".conv_spatial", a2, 0, dim)
        cat = e.tensor(x.n, x.h, x.w, 2 * half)
        h1, h2 = cat.channels(0, half), cat.channels(half, 2 * half)
        e.conv_group([a1, a2], [self._pack(p + ".conv1", [self._plain_part(p + ".conv1")], a1.c),
                                self._pack(p + ".conv2", [self._plain_part(p + ".conv2")], a2.c)], 1, 0, "none", outs=[h1, h2])
        self._rec(p + ".conv1", cat, 0, half)
        self._rec(p + ".conv2", cat, half, 2 * half)
        mm = e.channel_maxmean(cat)                        # [max, mean]
        key = (p + ".conv_squeeze", "swapped")
        if key not in self._packed:
            w = self.sd[p + ".conv_squeeze.weight"]
            self._packed[key] = e.pack_conv([(w[:, [1, 0]].contiguous(), torch.ones(w.shape[0]), self.sd[p + ".conv_squeeze.bias"])], mm.c)
        sig = e.conv(mm, self._packed[key], 1, 3, "sigmoid")
        self._rec(p + ".conv_squeeze", sig, 0, 2)
        mix = e.gate(h1, h2, sig)
        self._rec(p + ".mix", mix, 0, half)
        attn = self.plain(p + ".conv", mix)
        y = e.gate(x, attn, None, out=out)
        self._rec(p + ".out", y, 0, dim)
        return y

    def gating_variant(self, p: str) -> str:
        """Which member of the Patch_Conv_NonLocal family (new/Non_local_family.py:112-421) a checkpoint holds under p,
        told from its parameter names: the reference swaps them by editing Attention.__init__ (:258), never by a config key."""
        if self.has(p + ".conv_spatial.weight"):
            return "lsk"                                   # drone/models/lsk/LSK.py (darknet_lsk.py)
        if self.has(p + ".patchconv_lt_nonlocal.feat_patchconv_lt.conv.weight"):
            return "44"
        if self.has(p + ".attention_map.conv.weight"):
            return "adapt" if self.has(p + ".feat_patchconv_lt.conv.weight") else "adapt_new"
        return "new"

    def attention(self, p: str, x: TView, out: Optional[TView] = None) -> TView:
        """Attention (new/Non_local_family.py:254-272): proj_1 + exact GELU fused in one 1x1
        GEMM epilogue, gating unit, proj_2 with the shortcut add fused as the residual.  The gating unit is the
        reference's Patch_Conv_NonLocal_new unless the checkpoint holds one of its shape-preserving siblings."""
        t = self.plain(p + ".proj_1", x, act="gelu")
        kind = self.gating_variant(p + ".spatial_gating_unit")
        if kind == "adapt_new":
            t = self.patch_conv_nonlocal_adapt_new(p + ".spatial_gating_unit", t)
        elif kind == "new":
            t = self.patch_conv_nonlocal_new(p + ".spatial_gating_unit", t)
        elif kind == "lsk":
            t = self.lsk_block(p + ".spatial_gating_unit", t)
        else:
            raise ValueError("Attention cannot gate with Patch_Conv_NonLocal_%s: it halves the map (use it as a neck block)" % kind)
        return self.plain(p + ".proj_2", t, out=out, res=x)

    def spatial_attention(self, p: str, x: TView) -> TView:
        """SpatialAttention (new/Non_local_family.py:423-436) -> [n,h,w,8] view, channel 0 valid."""
        k = self.sd[p + ".conv.weight"].shape[-1]
        mm = self.e.channel_maxmean(x)
        self._rec(p + ".maxmean", mm, 0, 2)
        return self.plain(p + ".conv", mm, pad=k // 2, act="sigmoid")

    def identity_conv(self, p: str, x: TView, out: Optional[TView] = None) -> TView:
        k = self.sd[p + ".conv.weight"].shape[-1]
        return self.plain(p + ".conv", x, pad=k // 2, out=out)

    def stem_of_identity(self, stem: str, ident: str, x: TView) -> TView:
        """stems[k](Identity_Conv(x)) as ONE conv.  The identity conv is a plain k x k conv + bias (Identity_Conv.py:27-84) and
        at P5 its only reader is the head stem's 1x1 BaseConv (yolo_patch_nonlocal_plus.py:245-247 -> base/yolox.py:60): nothing
        non-linear lies between the two, so W[o, i] = sum_m W1[o, m] Wid[m, i] (float64 on the host) is a k x k conv straight
        to the stem's width and the identity conv's bias rides through the BN fold.  Same function, a quarter of the identity
        conv's multiplies at P5 (512 -> 128 instead of 512 -> 512 -> 128) and one launch fewer; P3 / P4 keep theirs (their
        identity convs also feed the stride-2 convs of the bottom-up path, whose zero padding a composed conv cannot express)."""
        w1, s1, b1 = self._bn_part(stem)
        wid, bid = self.sd[ident + ".conv.weight"].double(), self.sd[ident + ".conv.bias"].double()
        W1 = w1.double().reshape(w1.shape[0], -1)
        w = torch.einsum("om,mikl->oikl", W1, wid).float()
        b = (b1.double() + s1.double() * (W1 @ bid)).float()
        if self.composed is not None:
            self.composed[stem] = (w, s1, b)
        pk = self._pack(stem + "*" + ident, [(w, s1, b)], x.c)
        y = self.e.conv(x, pk, 1, wid.shape[-1] // 2, "silu")
        self._rec(stem, y, 0, w.shape[0])
        return y

    # ------------------------------------------------------------------ necks
    def pafpn(self, p: str, img: torch.Tensor, gl: bool, fold_p5: bool = False) -> List[TView]:
        """YOLOPAFPN (base/yolox.py:170-234) or its GL-fusion variant
        (block/non_local/yolo_patch_nonlocal_plus.py:180-247) when gl=True.  fold_p5: the caller's head is yolox_head,
        which may take P5_Identity into its stem (stem_of_identity); the third output is then C3_n4's."""
        sd = self.sd
        c3 = self.conv_out_channels(p + ".backbone.dark3.1.conv3")
        c4 = self.conv_out_channels(p + ".backbone.dark4.1.conv3")
        n, _, H, W = img.shape
        h3, w3, h4, w4, h5, w5 = H // 8, W // 8, H // 16, W // 16, H // 32, W // 32
        e = self.e
        extra = 1 if gl else 0
        # Patch_conv_feat1's linear channel_conv (2 mid -> c4) is read by C3_p4.conv1 / conv2 only (1x1 BaseConvs on the concat):
        # composed with them, the concat holds the block's [lr | tb] tensor instead and the channel_conv is never launched
        pc1 = p + ".Patch_conv_feat1"
        zc = 2 * self.conv_out_channels(pc1 + ".feat_patchconv_lt") if gl else 0
        fold1 = gl and (self.trace is None or self.composed is not None) and not os.environ.get("GLSDET_NO_PATCH_FOLD") and self.has(pc1 + ".channel_conv.bias") \
            and zc < c4 and not self.is_depthwise(p + ".C3_p4.conv1")
        cat_p4 = e.tensor(n, h4, w4, 2 * c4 + (zc if fold1 else extra * c4))      # [up(P5) | feat2 | (feat1_patch)]
        cat_p3 = e.tensor(n, h3, w3, 2 * c3)                # [up(P4) | feat1]
        cat_n3 = e.tensor(n, h4, w4, (2 + extra) * c3)      # [down(P3) | P4 | (feat2_patch)]
        cat_n4 = e.tensor(n, h5, w5, 2 * c4)                # [down(P4) | P5]
        f = self.darknet(p + ".backbone", img, {"dark3": cat_p3.channels(c3, 2 * c3),
                                                 "dark4": cat_p4.channels(c4, 2 * c4)})
        feat1, feat2, feat3 = f["dark3"], f["dark4"], f["dark5"]
        if gl:
            if fold1:
                for cv in ("conv1", "conv2"):
                    self.compose_1x1_input("%s.C3_p4.%s" % (p, cv), 2 * c4, 3 * c4, pc1 + ".channel_conv")
                self.patch_conv(pc1, feat1, 2, True, z_out=cat_p4.channels(2 * c4, 2 * c4 + zc))
            else:
                self.patch_conv(pc1, feat1, 2, True, out=cat_p4.channels(2 * c4, 3 * c4))
            self.patch_conv(p + ".Patch_conv_feat2", feat2, 1, False, out=cat_n3.channels(2 * c3, 3 * c3))
        P5 = self.cba(p + ".lateral_conv0", feat3, out=cat_n4.channels(c4, 2 * c4))
        e.resample(P5, 2, out=cat_p4.channels(0, c4))
        t = self.csp(p + ".C3_p4", cat_p4, False)
        P4 = self.cba(p + ".reduce_conv1", t, out=cat_n3.channels(c3, 2 * c3))
        e.resample(P4, 2, out=cat_p3.channels(0, c3))
        P3 = self.csp(p + ".C3_p3", cat_p3, False)
        if gl:
            P3 = self.identity_conv(p + ".P3_Identity", P3)
        self.cba(p + ".bu_conv2", P3, 2, out=cat_n3.channels(0, c3))
        P4o = self.csp(p + ".C3_n3", cat_n3, False)
        if gl:
            P4o = self.identity_conv(p + ".P4_Identity", P4o)
        self.cba(p + ".bu_conv1", P4o, 2, out=cat_n4.channels(0, c4))
        P5o = self.csp(p + ".C3_n4", cat_n4, False)
        self._p5_identity = None
        if gl:
            # (a trace holds every stored tensor, P5_Identity's output among them: tracing keeps the two-launch form)
            if fold_p5 and (self.trace is None or self.composed is not None) and not os.environ.get("GLSDET_NO_HEAD_FOLD"):
                self._p5_identity = p + ".P5_Identity"
            else:
                P5o = self.identity_conv(p + ".P5_Identity", P5o)
        self.features = f
        return [P3, P4o, P5o]

    # ------------------------------------------------------------------ head
    def _pred_parts(self, p: str, k: int, f: int, nc: int):
        """One 1x1 conv over [cls_feat | reg_feat] emitting cat([reg(4), obj(1), cls(nc)])."""
        w = torch.zeros(5 + nc, 2 * f, 1, 1)
        b = torch.zeros(5 + nc)
        w[0:4, f:] = self.sd["%s.reg_preds.%d.weight" % (p, k)]
        w[4:5, f:] = self.sd["%s.obj_preds.%d.weight" % (p, k)]
        w[5:, :f] = self.sd["%s.cls_preds.%d.weight" % (p, k)]
        b[0:4] = self.sd["%s.reg_preds.%d.bias" % (p, k)]
        b[4:5] = self.sd["%s.obj_preds.%d.bias" % (p, k)]
        b[5:] = self.sd["%s.cls_preds.%d.bias" % (p, k)]
        return w, torch.ones(5 + nc), b

    def yolox_head(self, p: str, feats: Sequence[TView]) -> List[TView]:
        """YOLOXHead.forward (base/yolox.py:46-92) -> per level fp32 [n,H,W,ceil8(5+nc)].
        Emitted stage by stage over the levels so that the small levels' convs of one shape class
        (second tower convs, predictors) can share a grouped launch."""
        L = len(feats)
        nc = self.sd["%s.cls_preds.0.weight" % p].shape[0]
        f = self.conv_out_channels("%s.stems.0" % p)
        ident = getattr(self, "_p5_identity", None)
        if ident is not None and (L != 3 or self.is_depthwise("%s.stems.2" % p)):
            raise ValueError("pafpn(fold_p5=True) left P5_Identity to a head that cannot take it in")
        self.stems = [self.stem_of_identity("%s.stems.%d" % (p, k), ident, x) if (ident is not None and k == 2)
                      else self.cba("%s.stems.%d" % (p, k), x) for k, x in enumerate(feats)]
        T = [self.cba(["%s.cls_convs.%d.0" % (p, k), "%s.reg_convs.%d.0" % (p, k)], s) for k, s in enumerate(self.stems)]
        U = [self.e.tensor(x.n, x.h, x.w, 2 * f) for x in feats]
        pairs = lambda ks: (["%s.%s_convs.%d.1" % (p, t, k) for k in ks for t in ("cls", "reg")],
                            [T[k].channels(i * f, (i + 1) * f) for k in ks for i in (0, 1)],
                            [U[k].channels(i * f, (i + 1) * f) for k in ks for i in (0, 1)])
        groups = [[0]] + ([[1, 2]] if L == 3 else [[k] for k in range(1, L)])     # level 0 is big enough alone
        for ks in groups:
            names, xs, outs = pairs(ks)
            self.cba_group(names, xs, outs=outs)
        packs = [self._pack("%s.preds.%d" % (p, k), [self._pred_parts(p, k, f, nc)], U[k].c) for k in range(L)]
        self.num_classes = nc
        return self.e.conv_group(U, packs, 1, 0, "none", out_dtype=F32)


    def cross_scale_head(self, p: str, feats: Sequence[TView]) -> List[TView]:
        """Cross-scale decoupled head (drone/models/lsk/yolox6.py:69-153 = new/yolox6.py).
        feats = (dark2, P3, P4, P5).  The cls tower of level k sees cat[x_k, down(finer level),
        up(coarser level)]: the stem writes x_k straight into that concat buffer, the two
        neighbours land in their slices (3x3 s1 + 3x3 s2 `up_convs`, nearest x2)."""
        e = self.e
        nc = self.sd["%s.cls_preds.0.weight" % p].shape[0]
        f = self.conv_out_channels("%s.stems.0" % p)
        feat0 = self.csp(p + ".csp_feat0", feats[0], False)
        lv, cats = [], []
        for k, x in enumerate(feats[1:]):
            parts = 3 if k < 2 else 2
            cat = e.tensor(x.n, x.h, x.w, parts * f)
            cats.append(cat)
            lv.append(self.cba("%s.stems.%d" % (p, k), x, out=cat.channels(0, f)))
        self.stems = lv
        outs = []
        for k in range(3):
            finer = feat0 if k == 0 else lv[k - 1]
            e.branch(1)
            t = self.cba("%s.up_convs.%d.0" % (p, k), finer)
            self.cba("%s.up_convs.%d.1" % (p, k), t, 2, out=cats[k].channels(f, 2 * f))
            if k < 2:
                e.branch(2)
                e.resample(lv[k + 1], 2, out=cats[k].channels(2 * f, 3 * f))
            e.branch(3)
            r = self.cba("%s.reg_convs.%d.0" % (p, k), lv[k])
            U = e.tensor(lv[k].n, lv[k].h, lv[k].w, 2 * f)
            self.cba("%s.reg_convs.%d.1" % (p, k), r, out=U.channels(f, 2 * f))
            e.branch(0)
            c = self.cba("%s.cls_convs.%d.0" % (p, k), cats[k])
            self.cba("%s.cls_convs.%d.1" % (p, k), c, out=U.channels(0, f))
            pk = self._pack("%s.preds.%d" % (p, k), [self._pred_parts(p, k, f, nc)], U.c)
            outs.append(e.conv(U, pk, 1, 0, "none", out_dtype=F32))
        self.num_classes = nc
        return outs


def build_forward(kind: str, eng: Engine, sd, img: torch.Tensor, trace: Optional[dict] = None, composed: Optional[dict] = None):
    """Emit the whole raw forward of a `kind` in {'base','gl','cross'} detector for the static input
    tensor `img` (NCHW fp32 on the device).  Returns (list of fp32 level views, num_classes).
    trace: dict filled with a snapshot of every stored tensor (eager emission only; parity tests).  composed: dict filled
    with the host-composed convs' (w, scale, bias); a trace then keeps the composed forms the benchmark runs."""
    b = NetBuilder(eng, sd)
    b.trace, b.composed = trace, composed
    if kind not in ("base", "gl", "cross"):
        raise ValueError("unknown detector kind %r" % kind)
    fold = kind == "gl" and not b.is_depthwise("head.stems.2")
    feats = b.pafpn("backbone", img, gl=(kind == "gl"), fold_p5=fold)
    if kind == "cross":
        outs = b.cross_scale_head("head", [b.features["dark2"]] + feats)
    else:
        outs = b.yolox_head("head", feats)
    return outs, b.num_classes, b.stems
