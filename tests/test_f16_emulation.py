"""The f16 mode (fp16 storage / fp32 accumulate, the benchmarked mode) against the fp16-storage
EMULATION of the oracle: `oracle.fp16_storage()` rounds the oracle's tensors to fp16 exactly where the
HIP path stores fp16 and nowhere else.  HIP-f16 vs emulation isolates KERNEL error (accumulation order,
v_exp / v_rcp SiLU, the re-associated non-local form); emulation vs the fp32 reference is the price of
fp16 storage on these nets (their conditioning), which no kernel can change.

CPU part: the emulation itself (identity when nothing is selected; rounding points; its distance from
the reference).  GPU part: every whole-model golden, logits and every stored tensor, per layer.
"""
import numpy as np
import pytest
import torch

from oracle import glsdet_oracle as O
from tests.helpers import block_case, model_case

MODELS = ["base_tiny_seed0", "base_tiny_seed1", "base_s_seed0", "gl_tiny_seed0", "gl_tiny_seed1", "gl_s_seed0",
          "base_nano_seed0", "base_nano_seed1", "gl_nano_seed0", "gl_nano_seed1"]

KERNEL_TOL = 2e-3        # HIP f16 vs emulation, x max|logit|


# --------------------------------------------------------------------------- CPU: the emulation
def test_emulation_with_nothing_selected_is_the_fp32_oracle(golden, shapes):
    meta, sd, x, outs, _ = model_case(golden, shapes, "gl_tiny_seed0")
    with O.fp16_storage(lambda name: False):
        got = O.yolox_gl_forward(sd, x)
    for g, w in zip(got, O.yolox_gl_forward(sd, x)):
        assert torch.equal(g, w)


def test_emulation_rounds_where_the_hip_path_stores(golden):
    """BaseConv output is fp16-representable, the Bottleneck shortcut is added BEFORE the one store,
    head logits / non-local conv_out weights stay fp32."""
    sd, x, want = block_case(golden, "bottleneck_add")
    with O.fp16_storage():
        O.TRACE = tr = {}
        y = O.bottleneck(sd, "m", x, True)
        O.TRACE = None
    assert torch.equal(y, y.half().float()) and not torch.equal(want, want.half().float())
    assert set(tr) == {"m.conv1", "m.conv2"} and torch.equal(tr["m.conv2"], y)
    # one rounding after the add: |y - (unrounded conv2 + x)| <= half an fp16 ulp of y
    with O.fp16_storage(lambda n: n != "m.conv2"):
        exact = O.bottleneck(sd, "m", x, True)
    assert float(((y - exact).abs() / exact.abs().clamp(min=2.0 ** -14)).max()) <= 2.0 ** -11
    sd, x, _ = block_case(golden, "nonlocal_c16")
    with O.fp16_storage():
        O.TRACE = tr = {}
        O.non_local_block(sd, "m", x.half().float())
        O.TRACE = None
    assert set(tr) == {"m.g", "m.theta", "m.phi", "m.conv_out"}


@pytest.mark.parametrize("tag", ["gl_s_seed0", "base_s_seed0"])
def test_emulation_distance_from_the_reference(golden, shapes, tag):
    """What fp16 storage costs on these random-weight nets (measured: max 0.043 / 0.076, rms 0.0076 / 0.0097
    x max|logit|; tools/f16_attribution.py prints where it comes from).  The HIP f16 path cannot be closer
    to the fp32 reference than this without storing something in more than 16 bits."""
    meta, sd, x, outs, _ = model_case(golden, shapes, tag)
    with O.fp16_storage():
        emu = O.FORWARDS[meta["model"]](sd, x)
    scale = max(float(w.abs().max()) for w in outs)
    d = torch.cat([(a - b).flatten() for a, b in zip(emu, outs)])
    mx, rms = float(d.abs().max()) / scale, float(d.pow(2).mean().sqrt()) / scale
    print("%s: emulation vs reference max %.4f rms %.5f (x max|logit| %.2f)" % (tag, mx, rms, scale))
    assert 5e-3 < mx < 0.12 and 1e-3 < rms < 0.015


# --------------------------------------------------------------------------- GPU: kernels vs the emulation
def _hip_trace(meta, sd, x, dtype):
    from glsdet_amd.engine import Engine
    from glsdet_amd.nets import build_forward
    eng = Engine(dtype)
    tr = {}
    outs, nc, _ = build_forward(meta["model"], eng, sd, x.cuda().float().contiguous(), trace=tr)   # eager: no plan
    torch.cuda.synchronize()
    return [o.to_nchw(5 + nc).cpu() for o in outs], tr


def _ulp16(t):
    """spacing of fp16 at |t| (normal range; 2^-24 below it)"""
    e = torch.floor(torch.log2(t.abs().clamp(min=2.0 ** -14)))
    return torch.pow(2.0, e - 10)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", MODELS)
def test_every_kernel_is_within_one_fp16_ulp_on_its_own_inputs(golden, shapes, tag):
    """KERNEL error, isolated from the conditioning of the net.  The emulation runs "teacher forced": after each
    of its ops has been compared with the tensor the HIP path stored at that point, the HIP tensor replaces the
    oracle's, so every oracle op consumes exactly the bytes the corresponding kernel consumed and nothing
    propagates.  Bar per stored element: one fp16 ulp (an fp32 summation-order difference can flip the final
    rounding) + 3e-5 x max|tensor| (fp32 cancellation noise under a value close to zero); the elements that
    differ at all stay under 10 % (measured 1-4 %); the fp32 logits, computed from forced inputs, agree to 5e-5."""
    meta, sd, x, outs, _ = model_case(golden, shapes, tag)
    got, tr = _hip_trace(meta, sd, x, "f16")
    ref = {}
    O.TRACE, O.FORCE = ref, tr
    try:
        with O.fp16_storage():
            emu = O.FORWARDS[meta["model"]](sd, x)
    finally:
        O.TRACE = O.FORCE = None
    missing = sorted(set(ref) - set(tr) - {"input"})
    assert not missing, "tensors the HIP trace does not cover: %s" % missing[:8]
    worst_ulp, worst_frac, n_el, n_diff = ("", 0.0), ("", 0.0), 0, 0
    for name, want in ref.items():
        if name == "input":
            continue
        have = tr[name]
        assert have.shape == want.shape, (name, have.shape, want.shape)
        d = (have - want).abs()
        tol = _ulp16(torch.maximum(have.abs(), want.abs())) + 3e-5 * float(want.abs().max())
        over = float((d / tol).max())
        frac = float((d > 0).float().mean())
        n_el += d.numel()
        n_diff += int((d > 0).sum())
        if over > worst_ulp[1]:
            worst_ulp = (name, over)
        if frac > worst_frac[1]:
            worst_frac = (name, frac)
    scale = max(float(w.abs().max()) for w in outs)
    dl = max(float((a - b).abs().max()) for a, b in zip(got, emu)) / scale
    print("%s: %d stored tensors, %.2f %% of %d elements differ from the forced emulation; worst element %.2f x tol (%s); "
          "most differing tensor %.1f %% (%s); forced logits max %.1e x max|logit|"
          % (tag, len(ref) - 1, 100.0 * n_diff / n_el, n_el, worst_ulp[1], worst_ulp[0], 100 * worst_frac[1], worst_frac[0], dl))
    assert worst_ulp[1] <= 1.0, worst_ulp
    assert n_diff <= 0.10 * n_el
    assert dl <= 5e-5


def _oracle_sd_with_composed_weights(sd, composed):
    """A state_dict under which the ORACLE computes the composed convs of the benchmarked GL-fusion plan with the composed
    weights themselves (so that they are rounded to fp16 as ONE matrix, as the HIP path packs them):
      Patch_conv_feat1.channel_conv -> [I ; 0]: its output is the block's [lr | tb] tensor padded with zero channels (exact),
          C3_p4.conv1 / conv2 read it through the composed columns W[:, c0:c1] Wlin (zero columns under the padding);
      P5_Identity -> the composed k x k conv to the stem's width in its first rows (no bias; its output is not a store point
          of the HIP plan and stays fp32: see `select`), head.stems.2 -> [I | 0] with the composed bias through its BN.
    BN of a composed consumer: gamma = the folded scale, beta = the composed bias, mean 0, var 1 - eps."""
    sd = dict(sd)

    def set_bn(p, s, b):
        sd[p + ".bn.weight"], sd[p + ".bn.bias"] = s.clone().float(), b.clone().float()
        sd[p + ".bn.running_mean"] = torch.zeros_like(s, dtype=torch.float32)
        sd[p + ".bn.running_var"] = torch.full_like(s, 1.0 - O.BN_EPS, dtype=torch.float32)

    lin = "backbone.Patch_conv_feat1.channel_conv"
    c4, zc = sd[lin + ".weight"].shape[:2]
    for cons in ("backbone.C3_p4.conv1", "backbone.C3_p4.conv2"):
        w, s_, b_ = composed[cons]
        co, cin = w.shape[:2]
        assert cin == 2 * c4 + zc and sd[cons + ".conv.weight"].shape[1] == 3 * c4
        sd[cons + ".conv.weight"] = torch.cat([w, torch.zeros(co, c4 - zc, 1, 1)], 1)
        set_bn(cons, s_, b_)
    eye = torch.zeros(c4, zc, 1, 1)
    eye[:zc, :, 0, 0] = torch.eye(zc)
    sd[lin + ".weight"], sd[lin + ".bias"] = eye, torch.zeros(c4)
    stem, ident = "head.stems.2", "backbone.P5_Identity"
    w, s_, b_ = composed[stem]
    f, cin, k, _ = w.shape
    full = torch.zeros_like(sd[ident + ".conv.weight"])
    assert full.shape[1:] == w.shape[1:] and sd[stem + ".conv.weight"].shape[1] == full.shape[0]
    full[:f] = w
    sd[ident + ".conv.weight"], sd[ident + ".conv.bias"] = full, torch.zeros(full.shape[0])
    eye = torch.zeros(f, full.shape[0], 1, 1)
    eye[:, :f, 0, 0] = torch.eye(f)
    sd[stem + ".conv.weight"] = eye
    set_bn(stem, s_, b_)
    return sd, (lambda name: name != ident + ".conv"), {lin, ident + ".conv"}


def test_oracle_on_the_composed_state_dict_equals_the_oracle_in_fp32(golden, shapes):
    """CPU: the construction the GPU test below relies on.  With the compositions done as NetBuilder does them (float64 on the
    host: compose_1x1_input, stem_of_identity) and nothing rounded, the oracle on the rewritten state_dict returns the
    reference's logits (fp32 re-association only)."""
    from glsdet_amd.nets import NetBuilder
    meta, sd, x, outs, _ = model_case(golden, shapes, "gl_tiny_seed0")
    b = NetBuilder(None, sd)
    b.composed = {}
    c4 = sd["backbone.Patch_conv_feat1.channel_conv.weight"].shape[0]
    for cons in ("backbone.C3_p4.conv1", "backbone.C3_p4.conv2"):
        b.compose_1x1_input(cons, 2 * c4, 3 * c4, "backbone.Patch_conv_feat1.channel_conv")
    w1, s1, b1 = b._bn_part("head.stems.2")
    wid, bid = sd["backbone.P5_Identity.conv.weight"].double(), sd["backbone.P5_Identity.conv.bias"].double()
    W1 = w1.double().reshape(w1.shape[0], -1)
    b.composed["head.stems.2"] = (torch.einsum("om,mikl->oikl", W1, wid).float(), s1, (b1.double() + s1.double() * (W1 @ bid)).float())
    sd2, select, gone = _oracle_sd_with_composed_weights(sd, b.composed)
    with torch.no_grad():
        got = O.FORWARDS["gl"](sd2, x)
    scale = max(float(w.abs().max()) for w in outs)
    err = max(float((g - w).abs().max()) for g, w in zip(got, outs)) / scale
    assert err <= 2e-5, err
    assert select("backbone.P5_Identity.conv.weight") and not select("backbone.P5_Identity.conv") and len(gone) == 2


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["gl_s_seed0", "gl_tiny_seed0", "gl_nano_seed0"])
def test_composed_convs_of_the_benchmarked_plan_are_within_one_fp16_ulp(golden, shapes, tag):
    """a trace used to switch the host-composed convs off (Patch_conv_feat1.channel_conv into C3_p4.conv1 /
    conv2, P5_Identity into head.stems.2), so the one-ulp test above verified a different op sequence than the captured plan.
    Here the trace KEEPS them (build_forward(composed=...)): the two tensors they remove are not in it, every other store
    point is, and the forced emulation runs on a state_dict that makes the oracle multiply by the very same composed weights
    (_oracle_sd_with_composed_weights).  Same bar: one fp16 ulp + 3e-5 x max|tensor| per stored element."""
    from glsdet_amd.engine import Engine
    from glsdet_amd.nets import build_forward
    meta, sd, x, outs, _ = model_case(golden, shapes, tag)
    assert meta["model"] == "gl"
    tr, comp = {}, {}
    hip, nc, _ = build_forward("gl", Engine("f16"), sd, x.cuda().float().contiguous(), trace=tr, composed=comp)
    torch.cuda.synchronize()
    got = [o.to_nchw(5 + nc).cpu() for o in hip]
    assert set(comp) == {"backbone.C3_p4.conv1", "backbone.C3_p4.conv2", "head.stems.2"}, sorted(comp)
    sd2, select, gone = _oracle_sd_with_composed_weights(sd, comp)
    assert not (gone & set(tr)), "the composed-away tensors must not exist in the HIP trace"
    ref = {}
    O.TRACE, O.FORCE = ref, tr
    try:
        with torch.no_grad(), O.fp16_storage(select):
            emu = O.FORWARDS["gl"](sd2, x)
    finally:
        O.TRACE = O.FORCE = None
    missing = sorted(set(ref) - set(tr) - {"input"} - gone)
    assert not missing, "tensors the HIP trace does not cover: %s" % missing[:8]
    worst, n_el, n_diff = ("", 0.0), 0, 0
    per = {}
    for name, want in ref.items():
        if name == "input" or name in gone:
            continue
        have = tr[name]
        assert have.shape == want.shape, (name, have.shape, want.shape)
        d = (have - want).abs()
        tol = _ulp16(torch.maximum(have.abs(), want.abs())) + 3e-5 * float(want.abs().max())
        over = float((d / tol).max())
        per[name] = over
        n_el += d.numel()
        n_diff += int((d > 0).sum())
        worst = max(worst, (name, over), key=lambda t: t[1])
    scale = max(float(w.abs().max()) for w in outs)
    dl = max(float((a - b).abs().max()) for a, b in zip(got, emu)) / scale
    print("%s composed plan: %d stored tensors, %.2f %% of %d elements differ; worst %.2f x tol (%s); composed convs: %s; "
          "forced logits max %.1e x max|logit|"
          % (tag, len(ref) - 1 - len(gone), 100.0 * n_diff / n_el, n_el, worst[1], worst[0],
             ", ".join("%s %.2f" % (k.split(".", 1)[1], per[k]) for k in sorted(comp)), dl))
    assert worst[1] <= 1.0, worst
    assert n_diff <= 0.10 * n_el
    assert dl <= 5e-5


@pytest.mark.gpu
@pytest.mark.parametrize("tag", MODELS)
def test_f16_mode_is_as_close_to_the_reference_as_fp16_storage_allows(golden, shapes, tag):
    """Free running (no forcing): HIP f16 vs the fp32 reference next to emulation vs the fp32 reference.  The
    emulation has perfect kernels, so its distance is the price of 16-bit storage on this net; the HIP path may
    not be further than 1.5 x that in rms (measured 0.8-1.3 x: the two differ by independent rounding flips,
    each a full fp16 ulp that the net then amplifies like any other perturbation -- tools/variant_check.py shows
    two bitwise-different CORRECT accumulation orders 4e-2 x max|logit| apart at the benchmark size)."""
    meta, sd, x, outs, _ = model_case(golden, shapes, tag)
    got, _ = _hip_trace(meta, sd, x, "f16")
    with O.fp16_storage():
        emu = O.FORWARDS[meta["model"]](sd, x)
    scale = max(float(w.abs().max()) for w in outs)
    cat = lambda a, b: torch.cat([(p - q).flatten() for p, q in zip(a, b)])
    r, e, d = cat(got, outs), cat(emu, outs), cat(got, emu)
    rms = lambda t: float(t.pow(2).mean().sqrt()) / scale
    print("%s: HIP f16 vs reference max %.4f rms %.5f | emulation vs reference max %.4f rms %.5f | HIP f16 vs emulation "
          "max %.4f rms %.5f (x max|logit| = %.2f)" % (tag, float(r.abs().max()) / scale, rms(r), float(e.abs().max()) / scale,
                                                      rms(e), float(d.abs().max()) / scale, rms(d), scale))
    assert rms(r) <= 1.5 * rms(e) + 1e-4
    assert float(r.abs().max()) / scale <= 2.0 * float(e.abs().max()) / scale + 1e-3
