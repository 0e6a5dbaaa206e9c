"""SURVEY section 8a row A6': the GL attention variants of drone/models/new/Non_local_family.py
(Patch_Conv_NonLocal_new, Attention, SpatialAttention) and the attention backbone
new/darknet_att.py.  Goldens come from the reference classes themselves
(tests/golden/make_golden.py::attention_cases -> attention_golden.npz).

CPU part: the oracle restatement against those goldens.  GPU part: the HIP lowering
(NetBuilder.attention / patch_conv_nonlocal_new / spatial_attention / darknet with lsk*)
against the same goldens, through the C ABI.  Tolerances as in test_hip_model.py:
f32 5e-5 * max(1,|ref|) per block, f16 2e-2 * max(1,|ref|)."""
import os

import pytest
import torch

from oracle import glsdet_oracle as O
from tests.helpers import block_case


def _dark(sd, x):
    f = O.csp_darknet_att(sd, "m.backbone", x)
    return torch.cat([f[k].flatten(1) for k in ("dark2", "dark3", "dark4", "dark5")], 1)


ORACLE = {
    "att_pcnl_new_nonlinear": lambda sd, x: O.patch_conv_nonlocal_new(sd, "m", x),
    "att_pcnl_new_linear": lambda sd, x: O.patch_conv_nonlocal_new(sd, "m", x),
    "att_attention_c32": lambda sd, x: O.attention(sd, "m", x),
    "att_attention_c48_odd": lambda sd, x: O.attention(sd, "m", x),
    "att_spatial_attention": lambda sd, x: O.spatial_attention(sd, "m", x),
    "att_pcnl_44": lambda sd, x: O.patch_conv_nonlocal_44(sd, "m", x),
    "att_pcnl_44_odd": lambda sd, x: O.patch_conv_nonlocal_44(sd, "m", x),
    "att_pcnl_adapt": lambda sd, x: O.patch_conv_nonlocal_adapt(sd, "m", x),
    "att_pcnl_adapt_nonlinear": lambda sd, x: O.patch_conv_nonlocal_adapt(sd, "m", x),
    "att_pcnl_adapt_new": lambda sd, x: O.patch_conv_nonlocal_adapt_new(sd, "m", x),
    "att_pcnl_adapt_new_linear": lambda sd, x: O.patch_conv_nonlocal_adapt_new(sd, "m", x),
    "att_darknet_tiny": _dark,
    "att_lskblock_c32": lambda sd, x: O.lsk_block(sd, "m", x),
    "att_lsk_attention_c48_odd": lambda sd, x: O.attention(sd, "m", x),
    "att_lsk_darknet_tiny": _dark,
}


def _err(a, b):
    return float((a - b).abs().max()) / max(1.0, float(b.abs().max()))


@pytest.mark.parametrize("tag", sorted(ORACLE))
def test_oracle_matches_reference(att_golden, tag):
    sd, x, want = block_case(att_golden, tag)
    got = ORACLE[tag](sd, x)
    assert got.shape == want.shape
    assert _err(got, want) <= 5e-5


def test_every_attention_golden_is_covered(att_golden):
    assert {k.split("/")[1] for k in att_golden.files if k.startswith("block/")} == set(ORACLE)


# ----------------------------------------------------------------------------------- HIP
@pytest.fixture(scope="module")
def engines():
    from glsdet_amd.engine import Engine
    return {"f32": Engine("f32"), "f16": Engine("f16")}


def _hip_dark(b, eng, x):
    f = b.darknet("m.backbone", x.cuda(), {})
    return torch.cat([f[k].to_nchw(f[k].c).flatten(1) for k in ("dark2", "dark3", "dark4", "dark5")], 1)


HIP = {
    "att_pcnl_new_nonlinear": lambda b, x: b.patch_conv_nonlocal_new("m", x),
    "att_pcnl_new_linear": lambda b, x: b.patch_conv_nonlocal_new("m", x),
    "att_attention_c32": lambda b, x: b.attention("m", x),
    "att_attention_c48_odd": lambda b, x: b.attention("m", x),
    "att_spatial_attention": lambda b, x: b.spatial_attention("m", x),
    "att_pcnl_44": lambda b, x: b.patch_conv_nonlocal_44("m", x),
    "att_pcnl_44_odd": lambda b, x: b.patch_conv_nonlocal_44("m", x),
    "att_pcnl_adapt_new": lambda b, x: b.patch_conv_nonlocal_adapt_new("m", x),
    "att_pcnl_adapt_new_linear": lambda b, x: b.patch_conv_nonlocal_adapt_new("m", x),
    "att_pcnl_adapt": lambda b, x: b.patch_conv_nonlocal_adapt("m", x),
    "att_pcnl_adapt_nonlinear": lambda b, x: b.patch_conv_nonlocal_adapt("m", x),
    "att_lskblock_c32": lambda b, x: b.lsk_block("m", x),
    "att_lsk_attention_c48_odd": lambda b, x: b.attention("m", x),
}


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["f32", "f16"])
@pytest.mark.parametrize("tag", sorted(HIP))
def test_hip_block_vs_reference_golden(engines, att_golden, mode, tag):
    from glsdet_amd.nets import NetBuilder
    from tests.test_hip_ops import _to_view
    eng = engines[mode]
    sd, x, want = block_case(att_golden, tag)
    out = HIP[tag](NetBuilder(eng, sd), _to_view(eng, x))
    torch.cuda.synchronize()
    got = out.to_nchw(want.shape[1]).cpu()
    tol = 5e-5 if mode == "f32" else 2e-2
    assert _err(got, want) <= tol, "%s/%s: %.3e" % (tag, mode, _err(got, want))


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["f32", "f16"])
@pytest.mark.parametrize("tag", ["att_darknet_tiny", "att_lsk_darknet_tiny"])
def test_hip_attention_backbone_vs_reference_golden(engines, att_golden, mode, tag):
    """new/darknet_att.py and lsk/darknet_lsk.py CSPDarknet (an Attention block after every stage: the quadrant
    non-local gating unit resp. the LSK block), all four stage outputs."""
    from glsdet_amd.nets import NetBuilder
    eng = engines[mode]
    sd, x, want = block_case(att_golden, tag)
    got = _hip_dark(NetBuilder(eng, sd), eng, x).cpu()
    torch.cuda.synchronize()
    assert got.shape == want.shape
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    noise = _err(want, _dark(sd64, x.double()).float())      # the reference's own fp32 rounding noise
    tol = max(1e-4, 2 * noise) if mode == "f32" else 5e-2
    print("att backbone %s: err %.3e (reference-vs-fp64 %.3e)" % (mode, _err(got, want), noise))
    assert _err(got, want) <= tol


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["f32", "f16"])
@pytest.mark.parametrize("c", [8, 64, 200])
def test_channel_maxmean(engines, mode, c):
    from tests.test_hip_ops import _to_view
    eng = engines[mode]
    x = O.synth_input((2, c, 9, 11), c)
    xr = x.half().float() if mode == "f16" else x
    out = eng.channel_maxmean(_to_view(eng, x, embed=(c + 16, 8)))
    torch.cuda.synchronize()
    got = out.to_nchw(8).cpu()
    assert float((got[:, 0] - xr.max(1)[0]).abs().max()) == 0.0
    assert float((got[:, 1] - xr.mean(1)).abs().max()) <= (1e-6 if mode == "f32" else 2e-3)
    assert float(got[:, 2:].abs().max()) == 0.0


def test_arch_table_equals_the_reference_attention_backbone(att_golden):
    """glsdet_amd.arch with attention_backbone=True lists new/darknet_att.py's state_dict
    (names, shapes, registration order) -- taken from the golden's recorded reference shapes."""
    from glsdet_amd.arch import state_dict_shapes
    from tests.helpers import meta_of
    ref = meta_of(att_golden, "block/att_darknet_tiny/meta")["shapes"]
    ours = [(k[len("backbone."):], list(v)) for k, v in state_dict_shapes("base", "tiny", 10, True).items()
            if k.startswith("backbone.backbone.")]
    assert ours == [(k, list(v)) for k, v in ref.items()]


def test_arch_table_equals_the_reference_lsk_backbone(att_golden):
    """... and with attention_backbone="lsk" lsk/darknet_lsk.py's (LSK.Attention: depthwise 5x5 / dilated 7x7 with bias)."""
    from glsdet_amd.arch import state_dict_shapes
    from tests.helpers import meta_of
    ref = meta_of(att_golden, "block/att_lsk_darknet_tiny/meta")["shapes"]
    ours = [(k[len("backbone."):], list(v)) for k, v in state_dict_shapes("cross", "tiny", 10, "lsk").items()
            if k.startswith("backbone.backbone.")]
    assert ours == [(k, list(v)) for k, v in ref.items()]


def test_drone_surface_has_the_lsk_detector_twins():
    import importlib
    import sys
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "glsdet_amd", "drone")
    sys.path.insert(0, root)
    try:
        for name in ("models.lsk.yolox6", "models.lsk.yolox6_lsk"):
            for k in [m for m in sys.modules if m == "models" or m.startswith("models.")]:
                del sys.modules[k]
            net = importlib.import_module(name).YoloBody(10, "tiny")
            keys = list(net.state_dict())
            assert "backbone.backbone.lsk2.spatial_gating_unit.conv_spatial.weight" in keys
            assert net.state_dict()["backbone.backbone.lsk3.spatial_gating_unit.conv0.weight"].shape[1:] == (1, 5, 5)
            assert any(k.startswith("head.csp_feat0.") for k in keys)          # the cross-scale head
    finally:
        sys.path.remove(root)
        for k in [m for m in sys.modules if m == "models" or m.startswith("models.")]:
            del sys.modules[k]


@pytest.mark.gpu
@pytest.mark.parametrize("kind,att", [("base", True), ("cross", "lsk")])
def test_detector_with_attention_backbone_vs_oracle(kind, att):
    """A whole detector whose backbone is the attention CSPDarknet (the lsk* keys switch it on): the quadrant non-local
    gating unit on the plain head, the LSK block under the cross-scale head (= drone/models/lsk/yolox6.py)."""
    from glsdet_amd.arch import state_dict_shapes
    from glsdet_amd.detector import HipDetector
    sd = O.synth_state_dict(state_dict_shapes(kind, "tiny", 10, att), 2)
    x = O.synth_input((1, 3, 96, 128), 5)
    want = O.FORWARDS[kind](sd, x)
    got = [g.cpu() for g in HipDetector(kind, sd, dtype="f32").forward_raw(x.cuda())]
    truth = [o.float() for o in O.FORWARDS[kind](
        {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}, x.double())]
    noise = max(_err(w, t) for w, t in zip(want, truth))
    for g, w in zip(got, want):
        assert g.shape == w.shape
        assert _err(g, w) <= max(1e-4, 2 * noise), (_err(g, w), noise)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["f32", "f16"])
@pytest.mark.parametrize("tag", ["att_pcnl_adapt_new", "att_pcnl_adapt_new_linear"])
def test_adaptive_split_is_found_on_the_device_and_survives_graph_capture(engines, att_golden, mode, tag):
    """The data-dependent split of Patch_Conv_NonLocal_adapt_new: the indices the device kernel leaves in device memory are
    the ones the reference's Python loops find (oracle.adapt_split on the same attention map), and the whole block --
    non-local windows, masked convs, select -- replays from a captured hipGraph with a DIFFERENT input (hence another
    split) without touching the host."""
    from glsdet_amd.nets import NetBuilder
    from tests.test_hip_ops import _to_view
    eng = engines[mode]
    sd, x, want = block_case(att_golden, tag)
    r = (lambda t: t.half().float()) if mode == "f16" else (lambda t: t)
    b = NetBuilder(eng, sd)
    xv = _to_view(eng, x)
    plan = eng.new_plan()
    with plan:
        out = b.patch_conv_nonlocal_adapt_new("m", xv)
    plan.run()
    torch.cuda.synchronize()
    assert b.last_split.cpu().tolist()[:3] == list(O.adapt_split(O.spatial_attention(sd, "m.attention_map", r(x))))
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        plan.capture(st)
    x2 = torch.roll(x, shifts=(7, 11), dims=(2, 3)) * 1.3            # another input: another split
    t = torch.zeros(x2.shape[0], x2.shape[2], x2.shape[3], xv.c)
    t[..., : x2.shape[1]] = x2.permute(0, 2, 3, 1)
    flat = xv.buf.view(torch.float16 if mode == "f16" else torch.float32)
    flat[: t.numel()] = t.flatten().to(flat.dtype).cuda()
    with torch.cuda.stream(st):
        plan.launch(st)
    st.synchronize()
    want2 = O.patch_conv_nonlocal_adapt_new(sd, "m", r(x2))
    split2 = list(O.adapt_split(O.spatial_attention(sd, "m.attention_map", r(x2))))
    assert b.last_split.cpu().tolist()[:3] == split2
    assert _err(out.to_nchw(want2.shape[1]).cpu(), want2) <= (5e-5 if mode == "f32" else 2e-2)


def test_gating_variant_is_told_from_the_checkpoint_keys(att_golden):
    """the reference swaps family members by editing Attention.__init__; a checkpoint says which by its names"""
    from glsdet_amd.nets import NetBuilder
    want = {"att_pcnl_44": "44", "att_pcnl_adapt": "adapt", "att_pcnl_adapt_new": "adapt_new", "att_pcnl_new_linear": "new",
            "att_lskblock_c32": "lsk"}
    for name, kind in want.items():
        sd, _, _ = block_case(att_golden, name)
        b = NetBuilder.__new__(NetBuilder)
        b.sd = sd
        assert b.gating_variant("m") == kind, name
