"""The mmdet flavour of the YOLOX path pinned by the reference's own code: tests/golden/yolox_mmdet_golden.npz holds what
yolox-ufp/mmdet's CSPDarknet + YOLOXPAFPN + YOLOXHead (loaded by file path, mmcv building blocks stood in: make_golden.py
yolox_mmdet_cases) produce on seeded data, together with their state_dict -- names, shapes, registration order."""
import os

import numpy as np
import pytest
import torch

from oracle import glsdet_oracle as O
from tests.helpers import block_case, meta_of

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = os.path.join(ROOT, "configs", "yolox", "yolox_s_visdrone.py")


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(ROOT, "tests", "golden", "yolox_mmdet_golden.npz"))


def _err(a, b):
    return float((a - b).abs().max()) / max(1.0, float(b.abs().max()))


def test_surface_state_dict_is_the_reference_mmdet_state_dict(gold):
    """YOLOX(CSPDarknet, YOLOXPAFPN, YOLOXHead) of glsdet_amd.mmdet_surface lists exactly the parameters and buffers the
    reference's mmdet modules register: same names, shapes and order -- so a reference checkpoint loads key for key."""
    from glsdet_amd.mmdet_surface import init_detector
    ref = meta_of(gold, "block/yolox_s_mmdet/meta")["shapes"]
    ours = [(k, list(v.shape)) for k, v in init_detector(CFG, device="cpu").state_dict().items()]
    assert ours == [(k, list(v)) for k, v in ref.items()]


def test_key_map_and_oracle_reproduce_the_reference_mmdet_forward(gold):
    """mmdet_to_drone_key (SURVEY 8a note) carries the mmdet parameters onto the drone network, whose oracle then gives the
    reference mmdet model's raw head outputs (per level reg | obj | cls)."""
    from glsdet_amd.mmdet_surface.models import mmdet_to_drone_key
    sd, x, want = block_case(gold, "yolox_s_mmdet")
    drone = {mmdet_to_drone_key(k[2:]): v for k, v in sd.items()}
    got = torch.cat([o.flatten(1) for o in O.FORWARDS["base"](drone, x)], 1)
    assert got.shape == want.shape
    d64 = {k: (v.double() if v.is_floating_point() else v) for k, v in drone.items()}
    noise = _err(want, torch.cat([o.flatten(1) for o in O.FORWARDS["base"](d64, x.double())], 1).float())
    assert _err(got, want) <= max(5e-5, 2 * noise), (_err(got, want), noise)


def test_oracle_decode_matches_the_reference_bbox_decode_and_priors(gold):
    pri = O.mlvl_point_priors([(12, 16), (6, 8), (3, 4)], [8, 16, 32])
    np.testing.assert_array_equal(pri.numpy(), gold["decode/priors"])
    got = O.mmdet_bbox_decode(pri, torch.from_numpy(gold["decode/preds"]))
    np.testing.assert_allclose(got.numpy(), gold["decode/boxes"], rtol=1e-6, atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["f32", "f16"])
def test_hip_surface_forward_vs_the_reference_mmdet_golden(gold, mode):
    """the HIP-backed surface model, loaded with the reference's parameter names, against the reference model's outputs"""
    from glsdet_amd.mmdet_surface import init_detector
    sd, x, want = block_case(gold, "yolox_s_mmdet")
    model = init_detector(CFG, cfg_options={"model.hip_dtype": mode})
    model.load_state_dict({k[2:]: v for k, v in sd.items()})
    got = torch.cat([t.cpu().flatten(1) for t in model._detector().forward_raw(x.cuda())], 1)
    assert got.shape == want.shape
    from glsdet_amd.mmdet_surface.models import mmdet_to_drone_key
    d64 = {mmdet_to_drone_key(k[2:]): (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    noise = _err(want, torch.cat([o.flatten(1) for o in O.FORWARDS["base"](d64, x.double())], 1).float())
    scale = max(1.0, float(want.abs().max()))
    rms = float((got - want).pow(2).mean().sqrt()) / scale
    print("mmdet yolox-s %s: max err %.3e rms %.3e (reference-vs-fp64 %.3e)" % (mode, _err(got, want), rms, noise))
    if mode == "f32":
        assert _err(got, want) <= max(1e-4, 2 * noise)
    else:
        assert rms <= 1.5e-2 and _err(got, want) <= 0.10


NANO = dict(backbone=dict(deepen_factor=0.33, widen_factor=0.25, use_depthwise=True),
            neck=dict(in_channels=[64, 128, 256], out_channels=64, num_csp_blocks=1, use_depthwise=True),
            bbox_head=dict(in_channels=64, feat_channels=64, use_depthwise=True))


def _nano_model(device="cpu", **opts):
    """the surface model of configs/yolox/yolox_nano_8x8_300e_coco.py:4-11 (use_depthwise=True in backbone, neck and head)"""
    from glsdet_amd.mmdet_surface import init_detector
    o = {"model.%s.%s" % (part, k): v for part, kv in NANO.items() for k, v in kv.items()}
    o.update(opts)
    return init_detector(CFG, device=device, cfg_options=o)


def test_depthwise_surface_state_dict_is_the_reference_mmdet_state_dict(gold):
    """use_depthwise=True (YOLOX-nano): DepthwiseSeparableConvModule's depthwise_conv / pointwise_conv
    parameters in the reference's names, shapes and registration order."""
    ref = meta_of(gold, "block/yolox_nano_mmdet/meta")["shapes"]
    ours = [(k, list(v.shape)) for k, v in _nano_model().state_dict().items()]
    assert ours == [(k, list(v)) for k, v in ref.items()]
    assert ("backbone.stage1.0.depthwise_conv.conv.weight", [16, 1, 3, 3]) in ours


def test_depthwise_key_map_and_oracle_reproduce_the_reference_mmdet_forward(gold):
    from glsdet_amd.mmdet_surface.models import mmdet_to_drone_key
    sd, x, want = block_case(gold, "yolox_nano_mmdet")
    drone = {mmdet_to_drone_key(k[2:]): v for k, v in sd.items()}
    assert "backbone.backbone.dark2.0.dconv.conv.weight" in drone and "head.cls_convs.0.1.pconv.bn.weight" in drone
    got = torch.cat([o.flatten(1) for o in O.FORWARDS["base"](drone, x)], 1)
    assert got.shape == want.shape
    d64 = {k: (v.double() if v.is_floating_point() else v) for k, v in drone.items()}
    noise = _err(want, torch.cat([o.flatten(1) for o in O.FORWARDS["base"](d64, x.double())], 1).float())
    assert _err(got, want) <= max(5e-5, 2 * noise), (_err(got, want), noise)


def test_the_reference_yolox_configs_build_on_the_surface():
    """every YOLOX config the reference ships (nano = depthwise included) resolves to surface classes"""
    import glob
    from glsdet_amd.mmdet_surface import init_detector
    cfgs = sorted(glob.glob("/root/reference/yolox-ufp/configs/yolox/yolox_*_8x8_300e_coco.py"))
    if not cfgs:
        pytest.skip("the reference tree is not present on this machine")
    for c in cfgs:
        m = init_detector(c, device="cpu")
        assert type(m).__name__ == "YOLOX" and len(m.state_dict()) > 300, c


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["f32", "f16"])
def test_hip_depthwise_surface_forward_vs_the_reference_mmdet_golden(gold, mode):
    sd, x, want = block_case(gold, "yolox_nano_mmdet")
    model = _nano_model(device="cuda:0", **{"model.hip_dtype": mode})
    model.load_state_dict({k[2:]: v for k, v in sd.items()})
    got = torch.cat([t.cpu().flatten(1) for t in model._detector().forward_raw(x.cuda())], 1)
    assert got.shape == want.shape
    from glsdet_amd.mmdet_surface.models import mmdet_to_drone_key
    d64 = {mmdet_to_drone_key(k[2:]): (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    noise = _err(want, torch.cat([o.flatten(1) for o in O.FORWARDS["base"](d64, x.double())], 1).float())
    scale = max(1.0, float(want.abs().max()))
    rms = float((got - want).pow(2).mean().sqrt()) / scale
    print("mmdet yolox-nano (depthwise) %s: max err %.3e rms %.3e (reference-vs-fp64 %.3e)" % (mode, _err(got, want), rms, noise))
    if mode == "f32":
        assert _err(got, want) <= max(1e-4, 2 * noise)
    else:
        # this depthwise net is ten times worse conditioned than yolox-s (fp32-vs-fp64 3e-4 against 3e-5), so fp16 STORAGE alone
        # costs several tenths of the logit range whoever computes it: the bar is the oracle's fp16-storage emulation of the same
        # graph (tests/test_f16_emulation.py): rms within 1.5 x, max within 2 x of what perfect kernels on fp16 tensors give
        drone = {mmdet_to_drone_key(k[2:]): v for k, v in sd.items()}
        with O.fp16_storage():
            emu = torch.cat([o.flatten(1) for o in O.FORWARDS["base"](drone, x)], 1)
        e_rms, e_max = float((emu - want).pow(2).mean().sqrt()) / scale, _err(emu, want)
        print("   fp16-storage emulation of the same net (no HIP code): max err %.3e rms %.3e" % (e_max, e_rms))
        assert rms <= 1.5 * e_rms + 1e-4 and _err(got, want) <= 2.0 * e_max + 1e-3
