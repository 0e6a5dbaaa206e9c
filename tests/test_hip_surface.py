"""GPU: the reference harness flow (drone/yolo.py:99-150) on the drop-in modules --
importlib YoloBody -> load_state_dict -> eval -> net(images) -> decode_outputs ->
non_max_suppression -- against the oracle's pipeline on the same seeded inputs."""
import importlib
import os
import sys

import numpy as np
import pytest
import torch

from oracle import glsdet_oracle as O
from tests.helpers import model_case

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture()
def drone_path(monkeypatch):
    monkeypatch.syspath_prepend(os.path.join(ROOT, "glsdet_amd", "drone"))
    for mod in list(sys.modules):
        if mod == "models" or mod.startswith("models."):
            monkeypatch.delitem(sys.modules, mod)


@pytest.mark.parametrize("cfg,tag", [("models/block/non_local/yolo_patch_nonlocal_plus.py", "gl_tiny_seed0"),
                                     ("models/base/yolox.py", "base_s_seed0")])
def test_harness_flow_matches_oracle(drone_path, golden, shapes, cfg, tag):
    meta, sd, x, outs, decoded = model_case(golden, shapes, tag)
    m = importlib.import_module(cfg[:-3].replace("/", "."))
    ub = importlib.import_module("models.core.utils_bbox")
    net = m.YoloBody(10, meta["phi"], dtype="f32")
    net.load_state_dict(sd)
    net = torch.nn.DataParallel(net.eval()).cuda()
    input_shape = meta["in_shape"][2:]
    image_shape = np.array([540, 1024])
    with torch.no_grad():
        outputs = net(x.cuda())
        assert [tuple(o.shape) for o in outputs] == [tuple(o.shape) for o in outs]
        dec = ub.decode_outputs(outputs, input_shape)
        res = ub.non_max_suppression(dec, 10, input_shape, image_shape, False, conf_thres=0.3, nms_thres=0.5)
    want_dec = O.decode_outputs(outs, input_shape)
    assert float(((dec.cpu() - want_dec).abs() / (want_dec.abs() + 1)).max()) < 1e-3
    # NMS on the oracle's decode of the HIP logits (same candidates either way)
    want = O.non_max_suppression(dec.cpu(), 10, input_shape, image_shape, False, 0.3, 0.5)
    assert len(res) == len(want)
    for a, b in zip(res, want):
        if b is None or len(b) == 0:
            assert a is None
            continue
        assert a.shape == b.shape
        np.testing.assert_array_equal(a[:, 6], b[:, 6])
        np.testing.assert_allclose(a[:, :4], b[:, :4], rtol=1e-4, atol=1e-2)    # pixels of a 1024x540 image
        np.testing.assert_allclose(a[:, 4:6], b[:, 4:6], rtol=1e-5, atol=1e-6)
    # plain tensors (no native handles) take the upload path and give the same decode
    dec2 = ub.decode_outputs([o.clone() for o in outputs], input_shape)
    assert torch.allclose(dec2, dec, rtol=0, atol=0)


def test_nothing_detected_returns_none(drone_path):
    ub = importlib.import_module("models.core.utils_bbox")
    pred = torch.zeros(2, 50, 15).cuda()
    res = ub.non_max_suppression(pred, 10, [64, 64], np.array([64, 64]), False, 0.5, 0.5)
    assert res == [None, None]
