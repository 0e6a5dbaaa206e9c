"""mmdet-flavour surface: registry / config loader / constructor validation / name map (CPU)
and the `model(return_loss=False, rescale=True, img=[..], img_metas=[[..]])` flow (GPU)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import glsdet_oracle as O
from tests.helpers import model_case

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_registry_build_semantics():
    from glsdet_amd.mmdet_surface import MODELS, BACKBONES, NECKS, HEADS, DETECTORS, Registry
    assert MODELS is BACKBONES is NECKS is HEADS is DETECTORS            # builder.py:7-15
    for name in ("CSPDarknet", "YOLOXPAFPN", "YOLOXHead", "YOLOX", "GLFusionPAFPN"):
        assert name in MODELS
    r = Registry("t")

    @r.register_module()
    class A:
        def __init__(self, x, y=2):
            self.x, self.y = x, y
    a = r.build(dict(type="A", x=1), default_args=dict(y=5, x=9))         # default_args never override
    assert (a.x, a.y) == (1, 5)
    with pytest.raises(KeyError):
        r.build(dict(type="Nope"))
    with pytest.raises(KeyError):
        r.build(dict(x=1))
    with pytest.raises(TypeError):
        r.build("A")
    with pytest.raises(KeyError):
        r.register_module(module=A)


def test_config_base_merge_and_delete(tmp_path):
    from glsdet_amd.mmdet_surface import Config
    (tmp_path / "base.py").write_text("model = dict(type='X', a=dict(p=1, q=2), b=3)\nlr = 0.1\n")
    (tmp_path / "child.py").write_text("_base_ = ['./base.py']\nmodel = dict(a=dict(_delete_=True, r=7), c=4)\n")
    cfg = Config.fromfile(str(tmp_path / "child.py"))
    assert cfg.model.type == "X" and cfg.model.b == 3 and cfg.model.c == 4 and cfg.lr == 0.1
    assert dict(cfg.model.a) == {"r": 7}
    cfg.merge_from_dict({"model.b": 9})
    assert cfg.model.b == 9


def test_repo_configs_build_and_match_reference_tables(shapes):
    from glsdet_amd.mmdet_surface import init_detector, mmdet_to_drone_key
    for path, tag in (("configs/yolox/yolox_s_visdrone.py", "base_s"),
                      ("configs/glsdet/yolox_s_glfusion_visdrone.py", "gl_s")):
        m = init_detector(os.path.join(ROOT, path))
        assert not m.training and m.cfg.model.type == "YOLOX"
        mapped = {mmdet_to_drone_key(k): tuple(v.shape) for k, v in m.state_dict().items()}
        assert set(mapped) == set(shapes[tag])
        assert all(tuple(shapes[tag][k]) == mapped[k] for k in mapped)
        assert "backbone.stage1.1.main_conv.conv.weight" in m.state_dict()
        assert "bbox_head.multi_level_conv_obj.2.bias" in m.state_dict()


@pytest.mark.skipif(not os.path.exists("/root/reference/yolox-ufp/configs/yolox/yolox_s_8x8_300e_coco.py"),
                    reason="reference checkout only exists in the build container")
def test_reference_yolox_config_file_loads_unchanged():
    from glsdet_amd.mmdet_surface import Config, build_detector
    cfg = Config.fromfile("/root/reference/yolox-ufp/configs/yolox/yolox_s_8x8_300e_coco.py")
    assert cfg.model.type == "YOLOX" and cfg.model.bbox_head.num_classes == 80
    assert cfg.model.test_cfg.nms.iou_threshold == 0.65 and cfg.optimizer.type == "SGD"
    m = build_detector(dict(cfg.model))
    assert m.bbox_head.num_classes == 80 and len(m.state_dict()) == 462


def test_constructor_validation_like_the_reference_tests():
    """ufp/tests/test_models/test_backbones/test_csp_darknet.py:11-17"""
    from glsdet_amd.mmdet_surface import CSPDarknet, YOLOXHead
    with pytest.raises(ValueError):
        CSPDarknet(frozen_stages=6)
    with pytest.raises(AssertionError):
        CSPDarknet(out_indices=[6])
    with pytest.raises(AssertionError):
        YOLOXHead(num_classes=4, in_channels=64, feat_channels=64, conv_bias="yes")


def test_bbox2result_format():
    from glsdet_amd.mmdet_surface import bbox2result
    d = np.array([[0, 0, 1, 1, .5, .8, 2], [1, 1, 2, 2, .9, .9, 0]], np.float32)
    r = bbox2result(d, 3)
    assert [x.shape for x in r] == [(1, 5), (0, 5), (1, 5)] and abs(r[2][0, 4] - 0.4) < 1e-7
    assert all(x.shape == (0, 5) for x in bbox2result(np.zeros((0, 7), np.float32), 3))


def _to_mmdet_sd(model, drone_sd):
    from glsdet_amd.mmdet_surface import mmdet_to_drone_key
    return {k: drone_sd[mmdet_to_drone_key(k)] for k in model.state_dict()}


@pytest.mark.gpu
@pytest.mark.parametrize("cfg,tag", [("configs/yolox/yolox_s_visdrone.py", "base_s_seed0"),
                                     ("configs/glsdet/yolox_s_glfusion_visdrone.py", "gl_s_seed0")])
def test_mmdet_flow_matches_oracle(golden, shapes, cfg, tag):
    from glsdet_amd.mmdet_surface import init_detector
    meta, sd, x, outs, _ = model_case(golden, shapes, tag)
    model = init_detector(os.path.join(ROOT, cfg), cfg_options={"model.hip_dtype": "f32",
                                                                "model.test_cfg.score_thr": 0.3})
    model.load_state_dict({"state_dict": _to_mmdet_sd(model, sd), "meta": {}})
    sf = np.array([0.625, 0.625, 0.625, 0.625], np.float32)
    metas = [[dict(img_shape=(128, 160, 3), ori_shape=(205, 256, 3), pad_shape=(128, 160, 3), scale_factor=sf, flip=False)
              for _ in range(x.shape[0])]]
    with torch.no_grad():
        res = model(return_loss=False, rescale=True, img=[x.cuda()], img_metas=metas)
    got_logits = [t.cpu() for t in model._detector().forward_raw(x.cuda())]
    want = O.mmdet_yolox_get_bboxes(got_logits, 10, [8, 16, 32], 0.3, 0.65, [sf] * x.shape[0])
    assert len(res) == x.shape[0] and all(len(r) == 10 for r in res)
    n_det = 0
    for ri, wi in zip(res, want):
        for a, b in zip(ri, wi):
            assert a.dtype == np.float32 and a.shape == b.shape
            np.testing.assert_allclose(a, b, rtol=1e-4, atol=1e-3)
            n_det += len(a)
    assert n_det > 0
    feats = model.extract_feat(x.cuda())
    assert [tuple(f.shape[1:]) for f in feats] == [(128, 16, 20), (128, 8, 10), (128, 4, 5)]
    with pytest.raises(NotImplementedError):
        model(return_loss=True, img=[x], img_metas=metas)
    with pytest.raises(TypeError):
        model(return_loss=False, img=x, img_metas=metas)
