"""Pins the CPU oracle (oracle/glsdet_oracle.py) against golden vectors produced by the
reference itself (tests/golden/make_golden.py, run in the build container with
/root/reference importable).  CPU-only; nothing here touches the HIP path."""
import numpy as np
import pytest
import torch

from oracle import glsdet_oracle as O
from tests.helpers import block_case, model_case, model_tags

TOL = 2e-5  # fp32 vs fp32, same op order up to conv algorithm choice


def _close(a, b, tol=TOL):
    scale = max(1.0, float(b.abs().max()))
    err = float((a - b).abs().max())
    assert a.shape == b.shape
    assert err <= tol * scale, "max abs err %.3e (scale %.2f)" % (err, scale)


BLOCKS = {
    "focus": lambda sd, x: O.focus(sd, "m", x),
    "spp": lambda sd, x: O.spp_bottleneck(sd, "m", x),
    "dwconv_k3_s2": lambda sd, x: O.dw_conv(sd, "m", x, 2),
    "bottleneck_add": lambda sd, x: O.bottleneck(sd, "m", x, True),
    "bottleneck_noadd": lambda sd, x: O.bottleneck(sd, "m", x, False),
    "csp_n2_shortcut": lambda sd, x: O.csp_layer(sd, "m", x, True),
    "csp_n1_noshortcut": lambda sd, x: O.csp_layer(sd, "m", x, False),
    "nonlocal_c16": lambda sd, x: O.non_local_block(sd, "m", x),
    "nonlocal_c32_inter16": lambda sd, x: O.non_local_block(sd, "m", x),
    "patch_conv_s1": lambda sd, x: O.patch_conv(sd, "m", x, 1, False),
    "patch_conv_nonlocal_s2": lambda sd, x: O.patch_conv(sd, "m", x, 2, True),
    "identity3": lambda sd, x: O.identity_conv(sd, "m", x),
    "identity5": lambda sd, x: O.identity_conv(sd, "m", x),
    "identity7": lambda sd, x: O.identity_conv(sd, "m", x),
}
for _k in (1, 3):
    for _s in (1, 2):
        for _a in ("silu", "relu", "lrelu"):
            BLOCKS["baseconv_k%d_s%d_%s" % (_k, _s, _a)] = (
                lambda sd, x, s=_s, a=_a: O.base_conv(sd, "m", x, s, a))


@pytest.mark.parametrize("tag", sorted(BLOCKS))
def test_block_matches_reference(golden, tag):
    sd, x, want = block_case(golden, tag)
    _close(BLOCKS[tag](sd, x), want)


def test_every_golden_block_is_covered(golden):
    tags = {k.split("/")[1] for k in golden.files if k.startswith("block/")}
    assert tags == set(BLOCKS)


def test_model_tags_present(golden):
    tags = model_tags(golden)
    assert "gl_s_seed0" in tags and "base_nano_seed0" in tags and len(tags) == 10


@pytest.mark.parametrize("tag", ["base_nano_seed0", "base_nano_seed1", "base_tiny_seed0",
                                 "base_tiny_seed1", "base_s_seed0", "gl_nano_seed0",
                                 "gl_nano_seed1", "gl_tiny_seed0", "gl_tiny_seed1", "gl_s_seed0"])
def test_model_matches_reference(golden, shapes, tag):
    meta, sd, x, outs, decoded = model_case(golden, shapes, tag)
    got = O.FORWARDS[meta["model"]](sd, x)
    for g, w in zip(got, outs):
        _close(g, w, 5e-5)
    dec = O.decode_outputs(got, meta["in_shape"][2:])
    # decoded wh = exp(logit)*stride can be huge: compare relatively per element
    rel = ((dec - decoded).abs() / (decoded.abs() + 1.0)).max()
    assert float(rel) < 1e-4


def test_decode_does_not_mutate_inputs(golden, shapes):
    meta, sd, x, outs, _ = model_case(golden, shapes, "base_nano_seed0")
    keep = [o.clone() for o in outs]
    O.decode_outputs(outs, meta["in_shape"][2:])
    for a, b in zip(outs, keep):
        assert torch.equal(a, b)


@pytest.mark.parametrize("lb", [0, 1])
def test_yolo_correct_boxes(golden, lb):
    got = O.yolo_correct_boxes(golden["correct_boxes/xy"].copy(), golden["correct_boxes/wh"].copy(),
                               [640, 640], np.array([540, 1024]), bool(lb))
    np.testing.assert_allclose(got, golden["correct_boxes/letterbox%d" % lb], rtol=1e-5, atol=1e-3)


# ---- NMS: no runnable reference (torchvision absent) -> property tests, parity unpinned
def _rand_boxes(rng, n):
    c = rng.uniform(0.1, 0.9, (n, 2))
    wh = rng.uniform(0.02, 0.3, (n, 2))
    return np.concatenate([c - wh / 2, c + wh / 2], 1).astype(np.float32)


def test_nms_properties():
    rng = np.random.default_rng(0)
    b = _rand_boxes(rng, 300)
    s = rng.uniform(0, 1, 300).astype(np.float32)
    l = rng.integers(0, 4, 300).astype(np.float32)
    keep = O.batched_nms(b, s, l, 0.5)
    assert len(set(keep.tolist())) == len(keep)
    assert np.all(np.diff(s[keep]) <= 0)                        # sorted by score desc
    assert np.array_equal(O.batched_nms(b[keep], s[keep], l[keep], 0.5), np.arange(len(keep)))  # idempotent

    def iou(a, bb):
        w = max(0, min(a[2], bb[2]) - max(a[0], bb[0]))
        h = max(0, min(a[3], bb[3]) - max(a[1], bb[1]))
        i = w * h
        return i / ((a[2] - a[0]) * (a[3] - a[1]) + (bb[2] - bb[0]) * (bb[3] - bb[1]) - i)
    for i in keep:                                              # survivors of a class don't overlap
        for j in keep:
            if i < j and l[i] == l[j]:
                assert iou(b[i], b[j]) <= 0.5 + 1e-6
    dropped = set(range(300)) - set(keep.tolist())
    for d in dropped:                                           # every dropped box has a better kept one
        assert any(l[k] == l[d] and s[k] >= s[d] and iou(b[k], b[d]) > 0.5 - 1e-6 for k in keep)


def test_nms_known_answer():
    # hand-computed: box1 overlaps box0 (IoU 0.68>0.5) same class -> dropped; box2 other class kept
    b = np.array([[0, 0, 10, 10], [1, 1, 11, 11], [1, 1, 11, 11], [20, 20, 30, 30]], np.float32)
    s = np.array([0.9, 0.8, 0.7, 0.6], np.float32)
    l = np.array([0, 0, 1, 0], np.float32)
    assert O.batched_nms(b, s, l, 0.5).tolist() == [0, 2, 3]
    assert O.batched_nms(b, s, l, 0.7).tolist() == [0, 1, 2, 3]
    assert O.batched_nms(b[:0], s[:0], l[:0], 0.5).tolist() == []


def test_non_max_suppression_shapes(golden, shapes):
    meta, sd, x, outs, decoded = model_case(golden, shapes, "gl_tiny_seed0")
    res = O.non_max_suppression(decoded, 10, meta["in_shape"][2:], np.array([540, 1024]), False, 0.3, 0.5)
    assert len(res) == 2
    for r in res:
        assert r is None or (r.ndim == 2 and r.shape[1] == 7)
    assert any(r is not None and len(r) > 0 for r in res)


@pytest.mark.parametrize("seed", [0, 1])
def test_cross_scale_head_matches_reference(golden, seed):
    """A7': oracle.cross_scale_head vs the reference head class (lsk/yolox6.py:7-153) itself"""
    import json
    meta = json.loads(bytes(golden["crosshead/seed%d/meta" % seed]).decode())
    sd = {"head." + k: torch.from_numpy(O.synth_tensor(k, tuple(s), seed)) for k, s in meta["shapes"].items()}
    feats = [O.synth_input(tuple(sh), seed + 300 + i) for i, sh in enumerate(meta["feat_shapes"])]
    outs = O.cross_scale_head(sd, "head", feats)
    for i, o in enumerate(outs):
        _close(o, torch.from_numpy(golden["crosshead/seed%d/out%d" % (seed, i)]))
