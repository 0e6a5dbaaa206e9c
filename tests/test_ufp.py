"""the second stage of UFPMP-Det.

Packing: `glsdet_amd.ufp.packing.unified_foreground_packing` against goldens of the reference's
own `UnifiedForegroundPacking` (tests/golden/make_golden.py::ufp_cases) -- identical chip lists."""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def ufp_boxes(trial):
    rng = np.random.default_rng([trial, 0x0F9])
    n = int(rng.integers(1, 60))
    W, H = int(rng.integers(400, 2000)), int(rng.integers(300, 1200))
    c = rng.uniform(0, 1, (n, 2)) * [W, H]
    wh = np.exp(rng.uniform(np.log(6), np.log(220), (n, 2)))
    b = np.concatenate([c - wh / 2, c + wh / 2], 1)
    b[:, 0::2] = np.clip(b[:, 0::2], 0, W - 1)
    b[:, 1::2] = np.clip(b[:, 1::2], 0, H - 1)
    return (b.astype(np.float32) if trial % 3 == 0 else b), W, H


@pytest.fixture(scope="module")
def ufp_golden():
    return np.load(os.path.join(ROOT, "tests", "golden", "ufp_golden.npz"))


@pytest.mark.parametrize("trial", range(24))
def test_packing_equals_the_reference(ufp_golden, trial):
    from glsdet_amd.ufp import unified_foreground_packing
    b, W, H = ufp_boxes(trial)
    chips, cw, ch = unified_foreground_packing(b.copy(), 1.5, [W, H])
    want = ufp_golden["ufp/%d/chips" % trial]
    assert np.array_equal(np.asarray(chips, np.float64).reshape(-1, 7), want)
    assert [cw, ch] == list(ufp_golden["ufp/%d/canvas" % trial])


def test_packing_properties():
    """chips do not overlap on the canvas, stay inside it, magnify by 1/2/4 and cover every input box."""
    from glsdet_amd.ufp import unified_foreground_packing
    for trial in (1, 5, 8, 13):
        b, W, H = ufp_boxes(trial)
        chips, cw, ch = unified_foreground_packing(b.copy(), 1.5, [W, H])
        rects = [(c[4], c[5], c[4] + c[2] * c[6], c[5] + c[3] * c[6]) for c in chips]
        for i, r in enumerate(rects):
            assert chips[i][6] in (1, 2, 4) and r[0] >= 0 and r[1] >= 0 and r[2] <= cw + 1e-9 and r[3] <= ch + 1e-9
            for q in rects[i + 1:]:
                assert min(r[2], q[2]) - max(r[0], q[0]) <= 1e-9 or min(r[3], q[3]) - max(r[1], q[1]) <= 1e-9
        for box in b:
            cx, cy = (box[0] + box[2]) / 2, (box[1] + box[3]) / 2
            assert any(c[0] <= cx <= c[0] + c[2] and c[1] <= cy <= c[1] + c[3] for c in chips)


# ---------------------------------------------------------------------------------- CPU: oracle pieces
def test_cv2_style_resize_restatement_known_answers():
    """2x INTER_LINEAR magnification of a 1x2 and a 2x2 uint8 image: the half-pixel weights are
    (1, .75/.25, .25/.75, 1) along each axis, borders clamp."""
    from oracle import ufp_oracle as U
    a = np.array([[[0, 0, 0], [100, 200, 40]]], np.uint8)
    r = U.cv2_resize_linear_u8(a, 4, 2)
    assert r.shape == (2, 4, 3)
    assert r[0, :, 0].tolist() == [0, 25, 75, 100] and r[1, :, 1].tolist() == [0, 50, 150, 200]
    b = np.array([[[0] * 3, [80] * 3], [[160] * 3, [240] * 3]], np.uint8)
    r = U.cv2_resize_linear_u8(b, 4, 4)[:, :, 0]
    assert r[0].tolist() == [0, 20, 60, 80] and r[3].tolist() == [160, 180, 220, 240] and r[1, 0] == 40 and r[2, 3] == 200
    assert np.array_equal(U.cv2_resize_linear_u8(b, 2, 2), b)


def test_py_cpu_nms_and_iof_known_answers():
    from oracle import ufp_oracle as U
    d = np.array([[0, 0, 9, 9, 0.9], [1, 1, 10, 10, 0.8], [20, 20, 29, 29, 0.7], [0, 0, 9, 9, 0.9]], np.float64)
    # '+1' areas: boxes 0/1 overlap 81/119 = 0.68 > 0.6 -> 1 dropped; the duplicate of box 0 (index 3) ranks first
    assert U.py_cpu_nms(d, 0.6) == [3, 2]
    assert U.py_cpu_nms(d, 0.7) == [3, 1, 2]
    assert U.compute_iof([0, 0, 10, 10], [5, 5, 100, 100]) == 0.25 and U.compute_iof([0, 0, 1, 1], [2, 2, 3, 3]) == 0.0


def _scene(trial=5, H=300, W=420):
    """source image + chips of a packing golden case rescaled into the image."""
    from glsdet_amd.ufp import unified_foreground_packing
    from tests.test_preprocess import synth_image
    rng = np.random.default_rng(trial)
    n = 14
    c = rng.uniform(0.1, 0.9, (n, 2)) * [W, H]
    wh = np.exp(rng.uniform(np.log(6), np.log(60), (n, 2)))
    b = np.concatenate([c - wh / 2, c + wh / 2], 1).astype(np.float32)
    b[:, 0::2] = np.clip(b[:, 0::2], 0, W - 1)
    b[:, 1::2] = np.clip(b[:, 1::2], 0, H - 1)
    chips, cw, ch = unified_foreground_packing(b.copy(), 1.5, [W, H])
    return synth_image((H, W), trial)[:, :, ::-1].copy(), chips, cw, ch


# ---------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("trial", [5, 6, 7])
def test_hip_mosaic_equals_the_restatement(trial):
    import torch
    from glsdet_amd.ufp import UfpSecondStage
    from oracle import ufp_oracle as U
    img, chips, cw, ch = _scene(trial)
    want = U.display_merge_result(img, chips, cw, ch)
    got = UfpSecondStage().mosaic(torch.from_numpy(img).cuda(), chips, cw, ch).cpu().numpy()
    assert got.shape == want.shape and {int(c[6]) for c in chips} <= {1, 2, 4}
    assert np.array_equal(got.astype(np.float64), want)          # integer arithmetic: exact
    assert want.max() > 0


@pytest.mark.gpu
@pytest.mark.parametrize("hw", [(270, 480), (765, 1360), (1080, 1920), (750, 1333), (97, 131)])
def test_hip_pipeline_input_uint8_frame_is_exact(hw):
    """First stage: a decoded uint8 frame goes through cv2's fixed-point resize (integer arithmetic: exact
    against the restatement), then the float normalisation (same operations: exact as well)."""
    import torch
    from glsdet_amd.ufp import UfpSecondStage
    from oracle import ufp_oracle as U
    img = np.random.default_rng(hw[0]).integers(0, 256, (hw[0], hw[1], 3), dtype=np.uint8)
    want, meta = U.mmdet_test_pipeline(img)
    got, m = UfpSecondStage().pipeline_input(torch.from_numpy(img).cuda())
    assert tuple(got.shape) == want.shape and m["img_shape"] == meta["img_shape"] and m["pad_shape"] == meta["pad_shape"]
    assert np.array_equal(got.cpu().numpy(), want)


@pytest.mark.gpu
def test_hip_pipeline_input_vs_restatement():
    import torch
    from glsdet_amd.ufp import UfpSecondStage
    from oracle import ufp_oracle as U
    img, chips, cw, ch = _scene(6)
    canvas = U.display_merge_result(img, chips, cw, ch)
    want, meta = U.mmdet_test_pipeline(canvas)
    got, m = UfpSecondStage().pipeline_input(torch.from_numpy(canvas.astype(np.float32)).cuda())
    assert tuple(got.shape) == want.shape and m["img_shape"] == meta["img_shape"] and m["pad_shape"] == meta["pad_shape"]
    assert np.array_equal(m["scale_factor"], meta["scale_factor"])
    assert np.abs(got.cpu().numpy() - want).max() <= 1e-5
    assert got.shape[2] % 32 == 0 and got.shape[3] % 32 == 0 and max(m["img_shape"][:2]) <= 1333


@pytest.mark.gpu
def test_hip_backmap_merge_vs_restatement():
    import torch
    from glsdet_amd.ufp import UfpSecondStage
    from oracle import ufp_oracle as U
    _, chips, cw, ch = _scene(7)
    rng = np.random.default_rng(3)
    rows = []
    for chip in chips:                                   # several detections inside every chip, clustered so that NMS bites
        ox, oy, w, h, nx, ny, s = [np.floor(v) for v in chip]
        for _ in range(6):
            bw, bh = rng.uniform(4, max(5, w * s / 3)), rng.uniform(4, max(5, h * s / 3))
            x, y = nx + rng.uniform(0, max(1, w * s - bw)), ny + rng.uniform(0, max(1, h * s - bh))
            for jitter in range(2):
                rows.append([x + jitter, y + jitter, x + bw + jitter, y + bh + jitter, rng.uniform(0.05, 1), 0, rng.integers(0, 3)])
    rows.append([cw * 0.9, -5, cw * 1.4, 30, 0.99, 0, 1])            # mostly outside every chip: dropped
    rows = np.asarray(rows, np.float32)
    rows[:, 5] = rows[:, 4]
    rows = rows[np.argsort(-rows[:, 4], kind="stable")]
    per_class = [rows[rows[:, 6] == c][:, :5] for c in range(3)]
    want = U.map_back_and_merge(per_class, chips, num_classes=3)
    dets = torch.zeros(len(rows) + 7, 7)
    dets[: len(rows)] = torch.from_numpy(rows)
    got = UfpSecondStage().merge(dets.cuda(), torch.tensor([len(rows), len(rows)], dtype=torch.int32).cuda(), chips, 3)
    assert sum(len(w) for w in want) > 10
    for c in range(3):
        assert len(got[c]) == len(want[c]), (c, len(got[c]), len(want[c]))
        np.testing.assert_allclose(got[c], want[c], rtol=1e-5, atol=1e-3)


@pytest.mark.gpu
def test_two_stage_pipeline_runs_end_to_end():
    """coarse GFL -> packing -> mosaic -> fine MPDet -> merge on a synthetic frame; every intermediate
    is checked against the restatement fed with the same upstream data."""
    import torch
    from glsdet_amd.resdet import HipGflDetector
    from glsdet_amd.synth import synth_input, synth_resdet_state_dict
    from glsdet_amd.ufp import TwoStagePipeline, UfpSecondStage, two_stage_detect
    from oracle import ufp_oracle as U
    from tests.test_preprocess import synth_image
    img = synth_image((270, 480), 2)[:, :, ::-1].copy()
    calib = synth_input((1, 3, 128, 160), 100)
    coarse = HipGflDetector("gfl", synth_resdet_state_dict("gfl", 0, calib), dtype="f16")
    fine = HipGflDetector("mpdet", synth_resdet_state_dict("mpdet", 1, calib), dtype="f16")
    stage = UfpSecondStage()

    def thr_for(det, x, keep):
        """random-init heads fire everywhere: pick the score threshold that lets ~keep pairs through"""
        cls, _ = det.forward_raw(x)
        p = torch.sigmoid(torch.cat([c.flatten() for c in cls]))
        return float(torch.topk(p, keep).values[-1])
    x1, _ = stage.pipeline_input(torch.from_numpy(img).cuda().contiguous())
    t1 = thr_for(coarse, x1, 60)
    merged, mid = two_stage_detect(coarse, fine, img, stage, dict(score_thr=t1, iou_thr=0.6, nms_pre=1000, max_per_img=40),
                                   dict(score_thr=0.999, iou_thr=0.6, nms_pre=1000, max_per_img=300))
    x2, _ = stage.pipeline_input(mid["canvas"])
    t2 = thr_for(fine, x2, 400)
    merged, mid = two_stage_detect(coarse, fine, img, stage, dict(score_thr=t1, iou_thr=0.6, nms_pre=1000, max_per_img=40),
                                   dict(score_thr=t2, iou_thr=0.6, nms_pre=1000, max_per_img=300))
    assert len(merged) == 10 and len(mid["chips"]) >= 1
    cw, ch = mid["canvas_wh"]
    assert np.array_equal(mid["canvas"].cpu().numpy().astype(np.float64), U.display_merge_result(img, mid["chips"], cw, ch))
    c = mid["fine_compiled"]
    k = int(c.nb["count"][0])
    rows = c.nb["dets"][0, :k].cpu().numpy()
    per_class = [rows[rows[:, 6] == cc][:, :5] for cc in range(10)]
    want = U.map_back_and_merge(per_class, mid["chips"])
    # float32 (device) vs float64 (restatement) IoF / IoU AT a threshold: a detection whose IoF with its chip is 0.9 to the last
    # bit is mapped back by one side only, and dropping it can un-suppress another row of its class (same count, other rows).
    # So: rows are matched as sets; over all classes at most three rows may be without a partner on either side.
    # (No range check: a fine-stage box may reach beyond the chip it is assigned to, and the reference maps it back through that
    #  chip's transform without clipping -- ufpmp_det_eval.py:282-300.)
    loose = 0
    for cc in range(10):
        g, w_ = np.asarray(merged[cc], np.float64).reshape(-1, 5), np.asarray(want[cc], np.float64).reshape(-1, 5)
        if len(g) == len(w_) and (len(g) == 0 or np.allclose(g, w_, rtol=1e-4, atol=1e-2)):
            continue
        assert abs(len(g) - len(w_)) <= 1, (cc, len(g), len(w_))
        if len(g) and len(w_):
            dist = np.abs(g[:, None, :] - w_[None, :, :]).max(-1)
            tol = 1e-2 + 1e-4 * max(np.abs(g).max(), np.abs(w_).max())
            loose += int((dist.min(1) > tol).sum()) + int((dist.min(0) > tol).sum())
        else:
            loose += len(g) + len(w_)
    assert loose <= 3, loose
    # the two-stream pipeline (coarse of frame i+1 beside fine of frame i) returns the same detections
    c1 = dict(score_thr=t1, iou_thr=0.6, nms_pre=1000, max_per_img=40)
    c2 = dict(score_thr=t2, iou_thr=0.6, nms_pre=1000, max_per_img=300)
    frames = [img, synth_image((270, 480), 3)[:, :, ::-1].copy(), img]
    seq = [two_stage_detect(coarse, fine, f, stage, c1, c2)[0] for f in frames]
    par = TwoStagePipeline(coarse, fine, stage, c1, c2).run(frames)
    for a, b in zip(seq, par):
        assert all(np.array_equal(x, y) for x, y in zip(a, b))
    # several lanes (each with private plan instances), and graph replay per input shape: same detections, in order
    for kw in (dict(workers=2), dict(workers=3, use_graph=True)):
        par = TwoStagePipeline(coarse, fine, stage, c1, c2, **kw).run(frames + frames)
        for a, b in zip(seq + seq, par):
            assert all(np.array_equal(x, y) for x, y in zip(a, b)), kw
