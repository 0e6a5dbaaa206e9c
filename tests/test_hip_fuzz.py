"""Randomised (fixed seed) sweep of glsdet_conv2d over shapes, strides, kernel sizes, views, residual
modes and EVERY kernel variant that accepts the problem, against torch's CPU conv: catches
tile-boundary / padding / variant-dispatch mistakes the hand-picked cases miss."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import glsdet_oracle as O
from tests.test_hip_ops import TOL, _cmp, _to_view

pytestmark = pytest.mark.gpu

GEMM_HINTS = list(range(16, 32))
GEO_HINTS = [(g << 8) | h for g in (1, 2) for h in (8, 9, 10, 11, 13)]       # ring kernels on 10 x 12 / 6 x 21 (10 x 24 / 6 x 42) pixel tiles       # persistent LDS-DMA 1x1 kernel (conv_gemm.hip), every variant
HINTS = [0, 1, 2, 3, 4, 5, 8, 9, 10, 11, 12, 13] + GEMM_HINTS + GEO_HINTS + [(128 << 16) | 128, (64 << 16) | 128, (64 << 16) | 64, (32 << 16) | 128,
         (64 << 16) | 64 | 0x8000, (128 << 16) | 128 | 0x8000, (128 << 16), (128 << 16) | 0x8000]


def _cases(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        k = int(rng.choice([1, 1, 3, 3, 3, 5, 7]))
        stride = int(rng.choice([1, 1, 1, 2]))
        cin = int(rng.choice([8, 16, 24, 32, 64, 72, 128, 136]))
        cout = int(rng.choice([8, 16, 40, 64, 96, 128, 136, 200]))
        h, w = int(rng.integers(k, 44)), int(rng.integers(k, 44))
        n_img = int(rng.choice([1, 2, 3]))
        act = str(rng.choice(["silu", "relu", "lrelu", "none", "gelu", "sigmoid"]))
        res = int(rng.choice([0, 0, 1, 2]))            # 0 none, 1 act-then-add, 2 add-then-act
        embed = bool(rng.integers(0, 2))
        out.append((n_img, cin, cout, k, stride, h, w, act, res, embed))
    return out


@pytest.fixture(scope="module")
def engines():
    from glsdet_amd.engine import Engine
    return {"f32": Engine("f32"), "f16": Engine("f16")}


def _run_case(engines, mode, case, hints):
    from glsdet_amd._lib import GlsdetError
    n_img, cin, cout, k, stride, h, w, act, res, embed = case
    eng = engines[mode]
    import zlib
    g = torch.Generator().manual_seed(zlib.crc32(repr(case).encode()))
    x = torch.randn(n_img, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) / np.sqrt(cin * k * k)
    scale, bias = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.3
    r = (lambda t: t.half().float()) if mode == "f16" else (lambda t: t)
    pre = F.conv2d(r(x), r(wt), None, stride, k // 2) * scale[None, :, None, None] + bias[None, :, None, None]
    act_fn = lambda t: F.gelu(t) if act == "gelu" else (torch.sigmoid(t) if act == "sigmoid" else O._act(t, act))
    rt = torch.randn(pre.shape, generator=g) if res else None
    ref = act_fn(pre) if res == 0 else (act_fn(pre) + r(rt) if res == 1 else act_fn(pre + r(rt)))
    pk = eng.pack_conv([(wt, scale, bias)], cin)
    ran = 0
    for hint in hints:
        xv = _to_view(eng, x, embed=(cin + 16, 8) if embed else None)
        rv = _to_view(eng, rt) if res else None
        try:
            out = eng.conv(xv, pk, stride, k // 2, act, res=rv, tile_hint=hint, res_first=(res == 2))
        except GlsdetError:
            assert hint != 0 and hint != 1, "the automatic choice and the generic kernel must accept every problem"
            continue
        torch.cuda.synchronize()
        _cmp(out.to_nchw(cout), ref, TOL[mode] * (2 if act in ("gelu", "sigmoid") else 1), "hint %x" % hint)
        ran += 1
    assert ran >= 2


@pytest.mark.parametrize("mode", ["f16", "f32"])
@pytest.mark.parametrize("case", _cases(40, 7), ids=lambda c: "n%d_ci%d_co%d_k%d_s%d_%dx%d_%s_r%d_e%d" % c)
def test_conv2d_random_problem_all_variants(engines, mode, case):
    _run_case(engines, mode, case, HINTS)


def _kxk_cases(n, seed):
    """stride-1 k x k problems with several channel chunks, several tiles per image and ragged borders: the
    territory of the halo kernels (register-staged, wave-private, LDS-DMA persistent, LDS-DMA weight ring)"""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        k = int(rng.choice([3, 3, 3, 5, 7]))
        cin = int(rng.choice([64, 128, 192, 256, 320]))
        cout = int(rng.choice([32, 64, 72, 128, 136, 256, 264]))
        h, w = int(rng.integers(9, 70)), int(rng.integers(17, 90))
        n_img = int(rng.choice([1, 2, 4]))
        act = str(rng.choice(["silu", "relu", "none"]))
        res = int(rng.choice([0, 1, 2]))
        embed = bool(rng.integers(0, 2))
        out.append((n_img, cin, cout, k, 1, h, w, act, res, embed))
    return out


@pytest.mark.parametrize("mode", ["f16", "f32"])
@pytest.mark.parametrize("case", _kxk_cases(16, 11), ids=lambda c: "n%d_ci%d_co%d_k%d_s%d_%dx%d_%s_r%d_e%d" % c)
def test_conv2d_random_kxk_problem_halo_family(engines, mode, case):
    _run_case(engines, mode, case, [0, 1, 2, 4, 5, 8, 9, 10, 11, 12, 13] + GEO_HINTS)


def _s2_cases(n, seed):
    """3x3 stride-2 problems (the downsampling convs: dark{2..5}.0, bu_conv, ResNet layer{2..4}.0.conv2, FPN extras):
    many tiles, several channel chunks, odd and even extents, ragged borders -- the de-interleaved-patch ring kernel."""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        cin = int(rng.choice([32, 64, 96, 128, 256]))
        cout = int(rng.choice([32, 64, 72, 128, 136, 256]))
        h, w = int(rng.integers(9, 110)), int(rng.integers(17, 130))
        n_img = int(rng.choice([1, 2, 3]))
        act = str(rng.choice(["silu", "relu", "none"]))
        res = int(rng.choice([0, 0, 1, 2]))
        embed = bool(rng.integers(0, 2))
        out.append((n_img, cin, cout, 3, 2, h, w, act, res, embed))
    return out


@pytest.mark.parametrize("mode", ["f16", "f32"])
@pytest.mark.parametrize("case", _s2_cases(14, 23), ids=lambda c: "n%d_ci%d_co%d_k%d_s%d_%dx%d_%s_r%d_e%d" % c)
def test_conv2d_random_stride2_problem_halo_family(engines, mode, case):
    _run_case(engines, mode, case, [0, 1, 10, 11, (64 << 16) | 128, 0x10a, 0x10b, 0x20a, 0x20b])


@pytest.mark.parametrize("case", _kxk_cases(6, 31) + _s2_cases(4, 37), ids=lambda c: "n%d_ci%d_co%d_k%d_s%d_%dx%d_%s_r%d_e%d" % c)
def test_ring_kernels_on_the_32x32x16_mfma_shape(engines, monkeypatch, case):
    """The fp16 ring kernels run v_mfma_f32_16x16x32_f16 by default (every f16 case above); GLSDET_NO_M16=1 selects the
    32x32x16 form they were first written on (read per launch): the same problems, every ring hint."""
    monkeypatch.setenv("GLSDET_NO_M16", "1")
    hints = [0, 1, 8, 9, 10, 11, 12, 13] + GEO_HINTS if case[4] == 1 else [0, 1, 10, 11, 0x10a, 0x10b, 0x20a, 0x20b]
    _run_case(engines, "f16", case, hints)


def _gemm_cases(n, seed):
    """1x1 problems for the persistent LDS-DMA kernel: ragged K (Cin not a whole 128-byte panel), one to ten panels, cout
    tiles with padding rows, pixel counts that end inside a tile, strided views, both residual orders"""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        cin = int(rng.choice([8, 16, 24, 32, 40, 64, 72, 96, 128, 136, 256, 264, 320, 512, 640]))
        cout = int(rng.choice([8, 16, 40, 64, 96, 128, 136, 192, 256, 264]))
        h, w = int(rng.integers(1, 60)), int(rng.integers(1, 60))
        n_img = int(rng.choice([1, 2, 3]))
        act = str(rng.choice(["silu", "relu", "lrelu", "none", "gelu"]))
        res = int(rng.choice([0, 0, 1, 2]))
        embed = bool(rng.integers(0, 2))
        out.append((n_img, cin, cout, 1, 1, h, w, act, res, embed))
    return out


@pytest.mark.parametrize("slots", [0, 3])
@pytest.mark.parametrize("mode", ["f16", "f32"])
@pytest.mark.parametrize("case", _gemm_cases(36, 23), ids=lambda c: "n%d_ci%d_co%d_k%d_s%d_%dx%d_%s_r%d_e%d" % c)
def test_conv1x1_persistent_dma_kernel_all_variants(engines, monkeypatch, mode, case, slots):
    """slots = 3: at most three workgroups per XCD (GLSDET_GEMM_SLOTS_PER_XCD), so that every workgroup walks many pixel
    tiles and the operand ring, the residual DMAs and the epilogues run across tile boundaries"""
    if slots:
        monkeypatch.setenv("GLSDET_GEMM_SLOTS_PER_XCD", str(slots))
    else:
        monkeypatch.delenv("GLSDET_GEMM_SLOTS_PER_XCD", raising=False)
    _run_case(engines, mode, case, [1] + GEMM_HINTS)


@pytest.mark.parametrize("slots", [0, 2])
@pytest.mark.parametrize("cin,cout,hw", [(256, 15, (25, 42)), (64, 15, (7, 9)), (136, 40, (33, 31)), (512, 200, (12, 11))])
def test_conv1x1_persistent_dma_kernel_fp32_output_of_fp16_input(engines, monkeypatch, cin, cout, hw, slots):
    """the predictor form: fp16 operands, fp32 logits (no rounding of the result), cout padded to 16 / 32"""
    from glsdet_amd._lib import F32, GlsdetError
    if slots:
        monkeypatch.setenv("GLSDET_GEMM_SLOTS_PER_XCD", str(slots))
    else:
        monkeypatch.delenv("GLSDET_GEMM_SLOTS_PER_XCD", raising=False)
    eng = engines["f16"]
    g = torch.Generator().manual_seed(cin + cout)
    x = torch.randn(3, cin, hw[0], hw[1], generator=g)
    wt = torch.randn(cout, cin, 1, 1, generator=g) / np.sqrt(cin)
    bias = torch.randn(cout, generator=g)
    ref = F.conv2d(x.half().float(), wt.half().float(), bias)
    pk = eng.pack_conv([(wt, torch.ones(cout), bias)], cin)
    ran = 0
    for hint in GEMM_HINTS:
        try:
            out = eng.conv(_to_view(eng, x), pk, 1, 0, "none", out_dtype=F32, tile_hint=hint)
        except GlsdetError:
            continue
        torch.cuda.synchronize()
        _cmp(out.to_nchw(cout), ref, 1e-3, "hint %d" % hint)
        ran += 1
    assert ran >= 2
