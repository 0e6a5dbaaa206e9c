"""GPU parity tests of the individual libglsdet_hip entry points against the CPU oracle
(fp32 torch).  Every call goes through the C ABI (glsdet_amd._lib via Engine).

Tolerances (stated per north_star: logits 1e-4 / boxes 1e-3 are met by the exact-f32
mode; the fp16-storage mode is held to fp16 rounding of the same arithmetic):
  f32 mode : |err| <= 2e-5 * max(1, |ref|_max)   per op
  f16 mode : |err| <= 4e-3 * max(1, |ref|_max)   per op  (2^-11 relative rounding of
             inputs/weights/outputs, fp32 accumulation)
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import glsdet_oracle as O

pytestmark = pytest.mark.gpu

TOL = {"f32": 2e-5, "f16": 4e-3}


@pytest.fixture(scope="module")
def engines():
    from glsdet_amd.engine import Engine
    return {"f32": Engine("f32"), "f16": Engine("f16")}


def _to_view(eng, x_nchw, pad_c=None, embed=None):
    """Upload NCHW fp32 -> NHWC view of the engine dtype.  embed=(C_total, c0): place the
    tensor as a channel slice of a wider buffer; a spatial border of 1 is added as well so
    the view is genuinely strided in n, h and w."""
    from glsdet_amd.engine import _TORCH_DT, TView
    n, c, h, w = x_nchw.shape
    dt = _TORCH_DT[eng.dt]
    if embed is None:
        v = eng.tensor(n, h, w, c)
        t = torch.zeros(n, h, w, v.c)
        t[..., :c] = x_nchw.permute(0, 2, 3, 1)
        v.buf.view(dt)[: t.numel()] = t.flatten().to(dt).to(eng.device)
        return v
    ctot, c0 = embed
    big = eng.tensor(n, h + 2, w + 2, ctot)
    t = torch.full((n, h + 2, w + 2, ctot), 7.0)          # poison around the window
    t[:, 1:-1, 1:-1, c0:c0 + c] = x_nchw.permute(0, 2, 3, 1)
    big.buf.view(dt)[: t.numel()] = t.flatten().to(dt).to(eng.device)
    return big.window(1, h + 1, 1, w + 1).channels(c0, c0 + c)


def _cmp(got, want, tol, what=""):
    got, want = got.cpu().float(), want.float()
    assert got.shape == want.shape, (got.shape, want.shape)
    scale = max(1.0, float(want.abs().max()))
    err = float((got - want).abs().max())
    assert err <= tol * scale, "%s: max abs err %.3e > %.1e * %.2f" % (what, err, tol, scale)


CONV_CASES = [
    # cin, cout, k, stride, act, H, W, residual, embed
    (16, 32, 3, 1, "silu", 20, 24, False, False),
    (16, 32, 3, 2, "relu", 20, 24, False, False),
    (16, 32, 1, 1, "lrelu", 20, 24, False, False),
    (16, 32, 1, 2, "none", 21, 23, False, False),
    (32, 64, 3, 1, "silu", 33, 17, True, False),
    (64, 128, 3, 1, "silu", 16, 16, True, True),
    (128, 128, 3, 1, "silu", 12, 20, False, True),
    (128, 256, 1, 1, "silu", 10, 12, False, False),
    (256, 192, 1, 1, "none", 9, 11, False, False),
    (24, 48, 3, 2, "silu", 16, 16, False, False),
    (16, 16, 5, 1, "none", 20, 24, False, False),
    (16, 16, 7, 1, "none", 20, 24, False, True),
    (384, 128, 1, 1, "silu", 8, 8, False, False),
    (64, 64, 3, 1, "silu", 40, 40, False, False),
    (8, 8, 3, 1, "none", 5, 5, False, False),
    (512, 512, 3, 1, "silu", 6, 7, False, False),
]


@pytest.mark.parametrize("mode", ["f32", "f16"])
@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: "ci%d_co%d_k%d_s%d_%s_%dx%d%s%s" % (
    c[0], c[1], c[2], c[3], c[4], c[5], c[6], "_res" if c[7] else "", "_emb" if c[8] else ""))
def test_conv2d(engines, mode, case):
    cin, cout, k, stride, act, H, W, use_res, embed = case
    eng = engines[mode]
    g = torch.Generator().manual_seed(cin * 1000 + cout + k)
    x = torch.randn(2, cin, H, W, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) / np.sqrt(cin * k * k)
    scale = torch.rand(cout, generator=g) + 0.5
    bias = torch.randn(cout, generator=g) * 0.3
    pad = (k - 1) // 2
    ref = F.conv2d(x, w, None, stride, pad) * scale[None, :, None, None] + bias[None, :, None, None]
    ref = O._act(ref, act)
    res = None
    if use_res:
        res = torch.randn(ref.shape, generator=g)
        ref = ref + res
    if mode == "f16":   # the kernel sees fp16-rounded operands; compare against the same
        xr, wr = x.half().float(), w.half().float()
        ref = O._act(F.conv2d(xr, wr, None, stride, pad) * scale[None, :, None, None] + bias[None, :, None, None], act)
        if use_res:
            ref = ref + res.half().float()
    xv = _to_view(eng, x, embed=(cin + 16, 8) if embed else None)
    rv = _to_view(eng, res) if use_res else None
    ov = None
    if embed:
        big = eng.tensor(2, ref.shape[2], ref.shape[3], cout + 24)
        ov = big.channels(16, 16 + cout)
    pk = eng.pack_conv([(w, scale, bias)], cin)
    out = eng.conv(xv, pk, stride, pad, act, out=ov, res=rv)
    torch.cuda.synchronize()
    _cmp(out.to_nchw(cout), ref, TOL[mode], "conv")


@pytest.mark.parametrize("mode", ["f32", "f16"])
def test_conv_fp32_output_and_fused_cout(engines, mode):
    """two weight sets fused along Cout; fp32 output view (the predictor path)"""
    from glsdet_amd._lib import F32
    eng = engines[mode]
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 64, 10, 12, generator=g)
    w1 = torch.randn(5, 64, 1, 1, generator=g) / 8
    w2 = torch.randn(10, 64, 1, 1, generator=g) / 8
    b1, b2 = torch.randn(5, generator=g), torch.randn(10, generator=g)
    xr = x.half().float() if mode == "f16" else x
    c = lambda w: (w.half().float() if mode == "f16" else w)
    ref = torch.cat((F.conv2d(xr, c(w1), b1), F.conv2d(xr, c(w2), b2)), 1)
    pk = eng.pack_conv([(w1, torch.ones(5), b1), (w2, torch.ones(10), b2)], 64)
    out = eng.conv(_to_view(eng, x), pk, 1, 0, "none", out_dtype=F32)
    torch.cuda.synchronize()
    assert out.dtype == F32 and out.c == 16
    _cmp(out.to_nchw(15), ref, TOL["f32"] if mode == "f32" else 1e-3, "pred conv")


@pytest.mark.parametrize("mode", ["f32", "f16"])
def test_focus_pack(engines, mode):
    eng = engines[mode]
    x = O.synth_input((2, 3, 32, 40), 3)
    tl, bl, tr, br = x[..., ::2, ::2], x[..., 1::2, ::2], x[..., ::2, 1::2], x[..., 1::2, 1::2]
    ref = torch.cat((tl, bl, tr, br), 1)
    out = eng.focus_pack(x.cuda())
    torch.cuda.synchronize()
    assert out.c == 16
    _cmp(out.to_nchw(12), ref if mode == "f32" else ref.half().float(), 1e-7, "focus")
    assert float(out.to_nchw()[:, 12:].abs().max()) == 0.0


@pytest.mark.parametrize("mode", ["f32", "f16"])
@pytest.mark.parametrize("k", [5, 9, 13])
def test_maxpool(engines, mode, k):
    eng = engines[mode]
    x = O.synth_input((2, 32, 20, 24), k)
    xr = x.half().float() if mode == "f16" else x
    out = eng.maxpool(_to_view(eng, x, embed=(64, 16)), k)
    torch.cuda.synchronize()
    _cmp(out.to_nchw(), F.max_pool2d(xr, k, 1, k // 2), 0.0, "maxpool")


@pytest.mark.parametrize("mode", ["f32", "f16"])
def test_maxpool_chain_equals_big_kernel(engines, mode):
    eng = engines[mode]
    x = O.synth_input((1, 8, 11, 7), 1)
    xr = x.half().float() if mode == "f16" else x
    a = eng.maxpool(_to_view(eng, x), 5)
    b = eng.maxpool(a, 5)
    c = eng.maxpool(b, 5)
    torch.cuda.synchronize()
    _cmp(b.to_nchw(), F.max_pool2d(xr, 9, 1, 4), 0.0, "5o5=9")
    _cmp(c.to_nchw(), F.max_pool2d(xr, 13, 1, 6), 0.0, "5o5o5=13")


@pytest.mark.parametrize("mode", ["f32", "f16"])
@pytest.mark.parametrize("factor", [1, 2])
def test_resample(engines, mode, factor):
    eng = engines[mode]
    x = O.synth_input((2, 24, 9, 11), 2)
    xr = x.half().float() if mode == "f16" else x
    big = eng.tensor(2, 9 * factor, 11 * factor, 64)
    out = eng.resample(_to_view(eng, x, embed=(40, 8)), factor, out=big.channels(24, 48))
    torch.cuda.synchronize()
    ref = F.interpolate(xr, scale_factor=factor, mode="nearest") if factor > 1 else xr
    _cmp(out.to_nchw(), ref, 0.0, "resample")
    assert float(big.to_nchw()[:, :24].abs().max()) == 0.0 and float(big.to_nchw()[:, 48:].abs().max()) == 0.0


@pytest.mark.parametrize("mode", ["f32", "f16"])
@pytest.mark.parametrize("count", [1, 5, 37])
def test_copy_many(engines, mode, count):
    """glsdet_copy_many: `count` independent strided copies of unequal extents (37 = two launches), bit exact, nothing
    written outside the destinations; malformed pairs refused on the host."""
    from glsdet_amd._lib import GlsdetError
    eng = engines[mode]
    rng = np.random.default_rng(count)
    srcs, dsts, refs, bigs = [], [], [], []
    for i in range(count):
        n, h, w = int(rng.integers(1, 3)), int(rng.integers(1, 14)), int(rng.integers(1, 19))
        x = O.synth_input((n, 24, h, w), 10 + i)
        refs.append(x.half().float() if mode == "f16" else x)
        srcs.append(_to_view(eng, x, embed=(40, 8)))
        big = eng.tensor(n, h + 2, w + 1, 64)
        bigs.append(big)
        dsts.append(big.window(1, h + 1, 0, w).channels(16, 40))
    eng.copy_many(srcs, dsts)
    torch.cuda.synchronize()
    for i in range(count):
        _cmp(dsts[i].to_nchw(), refs[i], 0.0, "copy_many %d" % i)
        full = bigs[i].to_nchw()
        h, w = refs[i].shape[2:]
        full[:, 16:40, 1:h + 1, 0:w] = 0
        assert float(full.abs().max()) == 0.0
    with pytest.raises(GlsdetError):
        eng.copy_many([srcs[0]], [eng.tensor(srcs[0].n, srcs[0].h + 1, srcs[0].w, 24)])
    with pytest.raises(GlsdetError):
        eng.copy_many([srcs[0]], [eng.tensor(srcs[0].n, srcs[0].h, srcs[0].w, 32)])


@pytest.mark.parametrize("mode", ["f32", "f16"])
@pytest.mark.parametrize("count", [1, 6, 35])
def test_transpose_many(engines, mode, count):
    """glsdet_transpose_many: window [1,h,w,C] -> matrix[c][pixel], bit exact; rows beyond C and columns beyond the
    pixels keep what they held (a preset ones row survives); too small a matrix is refused."""
    from glsdet_amd._lib import GlsdetError
    eng = engines[mode]
    rng = np.random.default_rng(100 + count)
    srcs, mats, refs = [], [], []
    tdt = torch.float16 if mode == "f16" else torch.float32
    for i in range(count):
        h, w = int(rng.integers(1, 15)), int(rng.integers(1, 23))
        c = 72
        x = O.synth_input((1, c, h, w), 30 + i)
        refs.append(x.half().float() if mode == "f16" else x)
        srcs.append(_to_view(eng, x, embed=(96, 8)))
        m = eng.matrix(c + 8, (h * w + 7) // 8 * 8 + 8)
        m.buf.view(tdt)[c * m.sw: c * m.sw + h * w] = 1.0                 # the ones row of the augmented operand
        mats.append(m)
    for s_, m in zip(srcs, mats):
        assert s_.c == 72
    eng.transpose_many(srcs, mats)
    torch.cuda.synchronize()
    for i in range(count):
        m, r = mats[i], refs[i]
        n = r.shape[2] * r.shape[3]
        full = m.buf.view(tdt)[: (72 + 8) * m.sw].view(72 + 8, m.sw).float().cpu()
        assert torch.equal(full[:72, :n], r[0].reshape(72, n)), i
        assert float(full[:72, n:].abs().max()) == 0.0
        assert torch.equal(full[72, :n], torch.ones(n)) and float(full[72, n:].abs().max()) == 0.0
        assert float(full[73:].abs().max()) == 0.0
    with pytest.raises(GlsdetError):
        eng.transpose_many([srcs[0]], [eng.matrix(64, srcs[0].h * srcs[0].w + 8)])
    wide = _to_view(eng, O.synth_input((1, 72, 4, 8), 1))
    with pytest.raises(GlsdetError):
        eng.transpose_many([wide], [eng.matrix(80, 24)])                    # 32 pixels need 32 columns


@pytest.mark.parametrize("mode", ["f32", "f16"])
@pytest.mark.parametrize("shape,ci", [((2, 16, 10, 12), 16), ((2, 32, 5, 7), 16), ((1, 64, 25, 42), 64),
                                      ((1, 384, 6, 9), 384), ((2, 200, 7, 5), 136), ((1, 72, 9, 9), 264)])
def test_nonlocal(engines, mode, shape, ci):
    from glsdet_amd.nets import NetBuilder
    eng = engines[mode]
    cx = shape[1]
    shapes = {"m.g.weight": (ci, cx, 1, 1), "m.g.bias": (ci,), "m.theta.weight": (ci, cx, 1, 1),
              "m.theta.bias": (ci,), "m.phi.weight": (ci, cx, 1, 1), "m.phi.bias": (ci,),
              "m.conv_out.weight": (cx, ci, 1, 1), "m.conv_out.bias": (cx,)}
    sd = O.synth_state_dict(shapes, 3)
    x = O.synth_input(shape, 4)
    ref = O.non_local_block(sd, "m", x)
    b = NetBuilder(eng, sd)
    xv = _to_view(eng, x)
    out = b.nonlocal_block("m", xv)
    torch.cuda.synchronize()
    _cmp(out.to_nchw(), ref, 1e-4 if mode == "f32" else 2e-2, "nonlocal")


def test_errors_are_reported_not_launched(engines):
    """bad operands -> negative return code -> GlsdetError (mirrors the reference's
    assertion / exception style, e.g. yolox_pafpn.py:125)"""
    from glsdet_amd._lib import GlsdetError
    eng = engines["f16"]
    x = eng.tensor(1, 8, 8, 16)
    w = torch.randn(16, 16, 3, 3)
    pk = eng.pack_conv([(w, torch.ones(16), torch.zeros(16))], 16)
    bad_out = eng.tensor(1, 7, 8, 16)
    with pytest.raises(GlsdetError):
        eng.conv(x, pk, 1, 1, "silu", out=bad_out)
    with pytest.raises(GlsdetError):
        eng.maxpool(x, 4)
    # a view that leaves its allocation must be refused on the host
    from glsdet_amd.engine import TView
    evil = TView(x.buf, 0, 1, 8, 8, 16, 8 * 8 * 16, 8 * 16 * 2, 16, x.dtype)
    with pytest.raises(GlsdetError):
        eng.maxpool(evil, 3)


HALO_CASES = [
    # cin, cout, k, H, W, residual, embed
    (64, 128, 3, 17, 23, True, True),
    (128, 128, 7, 20, 24, False, False),
    (128, 64, 5, 9, 33, False, True),
    (64, 64, 3, 8, 16, False, False),
    (256, 192, 3, 13, 10, True, False),
    (128, 256, 3, 40, 37, False, False),
]


@pytest.mark.parametrize("mode", ["f32", "f16"])
@pytest.mark.parametrize("hint", [1, 2, 4, 5, 8, 9, 10, 11],
                         ids=["generic", "halo", "halo_wave_private", "halo_co64", "halo_ring64", "halo_ring128",
                              "halo_ring64k64", "halo_ring128k64"])
@pytest.mark.parametrize("case", HALO_CASES, ids=lambda c: "ci%d_co%d_k%d_%dx%d" % c[:5])
def test_conv_halo_and_generic_kernels_agree_with_oracle(engines, mode, hint, case):
    """the stride-1 kxk halo kernel (hint 2) and the generic implicit GEMM (hint 1) on the
    same problems, incl. partial tiles, strided views, residual"""
    cin, cout, k, H, W, use_res, embed = case
    if hint in (4, 9, 11) and cout <= 64:
        pytest.skip("this halo variant needs the 128-row cout tile")
    eng = engines[mode]
    g = torch.Generator().manual_seed(cin + cout + k + H)
    x = torch.randn(2, cin, H, W, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) / np.sqrt(cin * k * k)
    scale = torch.rand(cout, generator=g) + 0.5
    bias = torch.randn(cout, generator=g) * 0.3
    r = (lambda t: t.half().float()) if mode == "f16" else (lambda t: t)
    ref = O._act(F.conv2d(r(x), r(w), None, 1, k // 2) * scale[None, :, None, None] + bias[None, :, None, None], "silu")
    res = None
    if use_res:
        res = torch.randn(ref.shape, generator=g)
        ref = ref + r(res)
    xv = _to_view(eng, x, embed=(cin + 16, 8) if embed else None)
    rv = _to_view(eng, res) if use_res else None
    ov = eng.tensor(2, H, W, cout + 24).channels(16, 16 + cout) if embed else None
    out = eng.conv(xv, eng.pack_conv([(w, scale, bias)], cin), 1, k // 2, "silu", out=ov, res=rv, tile_hint=hint)
    torch.cuda.synchronize()
    _cmp(out.to_nchw(cout), ref, TOL[mode], "conv halo/generic")


WS_CASES = [(128, 128, 20, 24, True, True), (64, 64, 33, 17, False, False), (256, 192, 9, 11, False, True),
            (128, 256, 30, 30, True, False), (64, 192, 7, 5, False, False)]


@pytest.mark.parametrize("mode", ["f32", "f16"])
@pytest.mark.parametrize("case", WS_CASES, ids=lambda c: "ci%d_co%d_%dx%d" % c[:4])
def test_conv1x1_weight_stationary(engines, mode, case):
    """persistent weight-stationary 1x1 kernel (tile_hint 3) incl. ragged tiles, views, residual"""
    cin, cout, H, W, use_res, embed = case
    if mode == "f32" and cin * 4 > 512:
        pytest.skip("resident weight tile limited to 512 bytes of K")
    eng = engines[mode]
    g = torch.Generator().manual_seed(cin + cout + H)
    x = torch.randn(3, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 1, 1, generator=g) / np.sqrt(cin)
    scale = torch.rand(cout, generator=g) + 0.5
    bias = torch.randn(cout, generator=g) * 0.3
    r = (lambda t: t.half().float()) if mode == "f16" else (lambda t: t)
    ref = O._act(F.conv2d(r(x), r(w)) * scale[None, :, None, None] + bias[None, :, None, None], "silu")
    res = None
    if use_res:
        res = torch.randn(ref.shape, generator=g)
        ref = ref + r(res)
    xv = _to_view(eng, x, embed=(cin + 16, 8) if embed else None)
    rv = _to_view(eng, res) if use_res else None
    ov = eng.tensor(3, H, W, cout + 24).channels(16, 16 + cout) if embed else None
    out = eng.conv(xv, eng.pack_conv([(w, scale, bias)], cin), 1, 0, "silu", out=ov, res=rv, tile_hint=3)
    torch.cuda.synchronize()
    _cmp(out.to_nchw(cout), ref, TOL[mode], "conv1x1 ws")


@pytest.mark.parametrize("mode", ["f32", "f16"])
@pytest.mark.parametrize("case", [
    # cin, cout, k, stride, act, [(H, W) per problem], residual
    (256, 128, 3, 1, "silu", [(13, 21), (12, 21), (13, 21), (12, 21)], False),
    (128, 64, 3, 2, "silu", [(50, 84), (50, 84), (50, 84), (50, 84)], False),
    (64, 192, 1, 1, "none", [(25, 42), (25, 41)], False),
    (128, 128, 3, 1, "relu", [(20, 24), (10, 12), (5, 6)], True),
    (16, 32, 3, 1, "lrelu", [(9, 9)], False),
])
def test_conv2d_multi_equals_separate_convs(engines, mode, case):
    """glsdet_conv2d_multi: n problems of one shape class, own weights / extents / residuals each."""
    eng = engines[mode]
    cin, cout, k, stride, act, sizes, use_res = case
    xs, packs, refs, ress = [], [], [], []
    for i, (h, w) in enumerate(sizes):
        g = torch.Generator().manual_seed(100 + i)
        x = torch.randn(2, cin, h, w, generator=g)
        wt = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
        sc, bi = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.2
        ho, wo = (h + 2 * (k // 2) - k) // stride + 1, (w + 2 * (k // 2) - k) // stride + 1
        res = torch.randn(2, cout, ho, wo, generator=g) if use_res else None
        r = lambda t: t.half().float() if mode == "f16" else t
        y = O._act(F.conv2d(r(x), r(wt), None, stride, k // 2) * sc.view(1, -1, 1, 1) + bi.view(1, -1, 1, 1), act)
        refs.append(y + r(res) if use_res else y)
        xs.append(_to_view(eng, x, embed=(cin + 16, 8)) if i % 2 else _to_view(eng, x))
        packs.append(eng.pack_conv([(wt, sc, bi)], cin))
        ress.append(_to_view(eng, res) if use_res else None)
    for hint in (0, (64 << 16) | 64, (128 << 16) | 128 | 0x8000):
        outs = eng.conv_multi(xs, packs, stride, k // 2, act, ress=ress, tile_hint=hint)
        torch.cuda.synchronize()
        for o, ref in zip(outs, refs):
            _cmp(o.to_nchw(cout), ref, TOL[mode], "conv multi hint %x" % hint)
    # the grouped ring kernel (conv_halo_ring_multi_kernel): 3x3 only, whole 128- (hints 8 / 9) or 64-byte (10 / 11) channel
    # chunks, stride 2 with 64-byte chunks only, 128-row cout tiles (9 / 11) for cout > 64; everything else is refused
    from glsdet_amd._lib import GlsdetError
    es = 2 if mode == "f16" else 4
    ran = 0
    for hint in (8, 9, 10, 11):
        applies = k == 3 and (cin * es) % (64 if hint >= 10 else 128) == 0 and (stride == 1 or hint >= 10) and \
            (hint in (8, 10) or cout > 64)
        if not applies:
            with pytest.raises(GlsdetError):
                eng.conv_multi(xs, packs, stride, k // 2, act, ress=ress, tile_hint=hint)
            continue
        outs = eng.conv_multi(xs, packs, stride, k // 2, act, ress=ress, tile_hint=hint)
        torch.cuda.synchronize()
        for o, ref in zip(outs, refs):
            _cmp(o.to_nchw(cout), ref, TOL[mode], "conv multi (grouped ring kernel) hint %d" % hint)
        ran += 1
    assert ran >= (1 if k == 3 and cin >= 32 else 0)


@pytest.mark.parametrize("mode", ["f32", "f16"])
@pytest.mark.parametrize("case", [
    # n problems, cin, cout, (H, W), residual, act
    (32, 64, 72, (9, 14), False, "none"),
    (19, 136, 40, (5, 23), True, "relu"),
    (9, 264, 200, (1, 70), False, "none"),
])
def test_conv2d_multi_batched_form_equals_separate_convs(engines, mode, case):
    """glsdet_conv2d_multi with 9..32 descriptors of ONE geometry (the batched form: one argument block + the operand
    addresses of each problem; the per-window GEMMs of the ResNet GL plug-in): every tile of the generic kernel, every
    problem bit-identical to its own glsdet_conv2d on the same tile; Engine.conv_many takes the form by itself and falls
    back to eight per launch when one problem's geometry differs; other hints and mixed geometries are refused."""
    from glsdet_amd._lib import GlsdetError
    eng = engines[mode]
    n, cin, cout, (h, w), use_res, act = case
    xs, packs, ress, refs = [], [], [], []
    r = (lambda t: t.half().float()) if mode == "f16" else (lambda t: t)
    for i in range(n):
        g = torch.Generator().manual_seed(500 + i)
        x = torch.randn(1, cin, h, w, generator=g)
        wt = torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5
        sc, bi = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.2
        res = torch.randn(1, cout, h, w, generator=g) if use_res else None
        y = O._act(F.conv2d(r(x), r(wt)) * sc.view(1, -1, 1, 1) + bi.view(1, -1, 1, 1), act)
        refs.append(y + r(res) if use_res else y)
        xs.append(_to_view(eng, x))
        packs.append(eng.pack_conv([(wt, sc, bi)], cin))
        ress.append(_to_view(eng, res) if use_res else None)
    for hint in (0, (64 << 16) | 64, (64 << 16) | 128, (128 << 16) | 128, (64 << 16) | 64 | 0x8000):
        outs = eng.conv_multi(xs, packs, 1, 0, act, ress=ress, tile_hint=hint)
        torch.cuda.synchronize()
        for i, o in enumerate(outs):
            _cmp(o.to_nchw(cout), refs[i], TOL[mode], "batched multi hint %x problem %d" % (hint, i))
            if hint and not (cout <= 64 and (hint >> 16) == 128):       # (a single conv refuses a tile that is mostly padding)
                one = eng.conv(xs[i], packs[i], 1, 0, act, res=ress[i], tile_hint=hint)
                torch.cuda.synchronize()
                assert torch.equal(o.to_nchw(cout), one.to_nchw(cout)), (hex(hint), i)
    outs = [eng.tensor(1, h, w, cout) for _ in range(n)]
    eng.conv_many(xs, packs, 1, 0, act, outs, ress=ress)
    torch.cuda.synchronize()
    for i, o in enumerate(outs):
        one = eng.conv(xs[i], packs[i], 1, 0, act, res=ress[i])
        torch.cuda.synchronize()
        assert float((o.to_nchw(cout) - one.to_nchw(cout)).abs().max()) <= TOL[mode] * max(1.0, float(one.to_nchw(cout).abs().max()))
    with pytest.raises(GlsdetError):
        eng.conv_multi(xs, packs, 1, 0, act, ress=ress, tile_hint=8)             # the grouped ring kernel takes eight at most
    odd = _to_view(eng, torch.randn(1, cin, h, w + 1))
    with pytest.raises(GlsdetError):
        eng.conv_multi(xs[:-1] + [odd], packs, 1, 0, act, ress=None if not use_res else ress[:-1] + [None])
    if not use_res:                                                              # conv_many: eight per launch then
        outs2 = [eng.tensor(1, h, w + (1 if i == n - 1 else 0), cout) for i in range(n)]
        eng.conv_many(xs[:-1] + [odd], packs, 1, 0, act, outs2)
        torch.cuda.synchronize()
        for i in (0, 3, n - 2):
            _cmp(outs2[i].to_nchw(cout), refs[i], TOL[mode], "conv_many fallback, problem %d" % i)


def test_tune_cache_keys_of_every_tuned_form_survive_the_json_file(tmp_path, monkeypatch):
    """The tuning table is keyed by flat tuples and stored as JSON (GLSDET_TUNE_CACHE): a second engine must load what the
    first one measured -- single convs, grouped launches, the batched form with and without a residual -- key for key."""
    from glsdet_amd.engine import Engine
    monkeypatch.setenv("GLSDET_TUNE_CACHE", str(tmp_path / "tune.json"))
    eng = Engine("f16", autotune=True)
    g = torch.Generator().manual_seed(3)
    xs = [_to_view(eng, torch.randn(1, 64, 6, 11, generator=g)) for _ in range(12)]
    pk = [eng.pack_conv([(torch.randn(40, 64, 1, 1, generator=g) / 8, torch.ones(40), torch.zeros(40))], 64) for _ in range(12)]
    rs = [_to_view(eng, torch.randn(1, 40, 6, 11, generator=g)) for _ in range(12)]
    for ress in (None, rs):
        eng.conv_many(xs, pk, 1, 0, "none", [eng.tensor(1, 6, 11, 40) for _ in range(12)], ress=ress)
    eng.conv_group(xs[:3], pk[:3], 1, 0, "relu")
    eng.conv(xs[0], pk[0], 1, 0, "relu")
    torch.cuda.synchronize()
    eng.save_tune_cache()
    assert sum(1 for k in eng._tuned if k[0] == "batch") >= 2
    import json
    with open(tmp_path / "tune.json") as f:
        loaded = {tuple(json.loads(k)): v for k, v in json.load(f)["f16"].items()}        # (Engine.__init__'s own parse)
    assert loaded == dict(eng._tuned) and len(loaded) >= 4
    Engine("f16", autotune=True)                       # and a second engine loads the file without complaint


@pytest.mark.parametrize("mode", ["f32", "f16"])
def test_nonlocal_multi_unequal_sets(engines, mode):
    """glsdet_nonlocal_multi: four quadrant windows of one tensor with different extents and weights."""
    from glsdet_amd.nets import NetBuilder
    eng = engines[mode]
    cx = ci = 32
    x = O.synth_input((2, cx, 13, 19), 9)
    shapes = {}
    names = ["q%d" % i for i in range(4)]
    for nm in names:
        shapes.update({nm + ".g.weight": (ci, cx, 1, 1), nm + ".g.bias": (ci,), nm + ".theta.weight": (ci, cx, 1, 1),
                       nm + ".theta.bias": (ci,), nm + ".phi.weight": (ci, cx, 1, 1), nm + ".phi.bias": (ci,),
                       nm + ".conv_out.weight": (cx, ci, 1, 1), nm + ".conv_out.bias": (cx,)})
    sd = O.synth_state_dict(shapes, 5)
    wins = [(0, 6, 0, 9), (6, 13, 0, 9), (0, 6, 9, 19), (6, 13, 9, 19)]
    ref = x.clone()
    for nm, (h0, h1, w0, w1) in zip(names, wins):
        ref[:, :, h0:h1, w0:w1] = O.non_local_block(sd, nm, x[:, :, h0:h1, w0:w1])
    xv = _to_view(eng, x)
    b = NetBuilder(eng, sd)
    b.nonlocal_blocks(names, [xv.window(h0, h1, w0, w1) for (h0, h1, w0, w1) in wins])
    torch.cuda.synchronize()
    _cmp(xv.to_nchw(), ref, 1e-4 if mode == "f32" else 2e-2, "nonlocal multi")


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["f32", "f16"])
@pytest.mark.parametrize("c,hw", [(32, (20, 24)), (256, (25, 42)), (8, (7, 50)), (64, (33, 17))])
def test_spp_pools_equal_three_max_pools(engines, mode, c, hw):
    """glsdet_spp_pools (5 / 9 / 13 in one launch, pool9 = pool5 o pool5, pool13 = pool5 o pool9 in LDS) vs
    F.max_pool2d(k, 1, k // 2) of darknet.py:29,35 and vs the chained single-pool launches: exact."""
    eng = engines[mode]
    x = torch.randn(2, c, hw[0], hw[1], generator=torch.Generator().manual_seed(c + hw[0]))
    if mode == "f16":
        x = x.half().float()
    cat = eng.tensor(2, hw[0], hw[1], 4 * c)
    xv = _to_view(eng, x)
    eng.resample(xv, 1, out=cat.channels(0, c))
    eng.spp_pools(cat.channels(0, c), cat.channels(c, 2 * c), cat.channels(2 * c, 3 * c), cat.channels(3 * c, 4 * c))
    torch.cuda.synchronize()
    got = cat.to_nchw().cpu()
    for i, k in enumerate((5, 9, 13)):
        want = F.max_pool2d(x, k, 1, k // 2)
        assert torch.equal(got[:, (i + 1) * c:(i + 2) * c], want), k
