"""Rows f.2 and f.4 pinned by fixtures of the REFERENCE's own pure-Python code (tests/golden/make_golden.py,
`merge_cases` / `eval_cases`): `compute_iof` and `py_cpu_nms` of ufp/ufpmp_det_eval.py:36-50,149-178 and the
vendored COCOeval of drone/models/core/cocoeval.py (`_prepare`, `computeIoU`'s ordering, `evaluateImg`, `accumulate`,
`summarize`).  Not pinned by them: the IoU arithmetic of pycocotools' compiled `_mask.iou` (restated in
cocoeval_oracle.bb_iou; the fixture generator routes the reference through that restatement).

CPU: oracle == fixture.  GPU: HIP == fixture."""
import copy
import json
import os

import numpy as np
import pytest

from oracle import cocoeval_oracle as CO
from oracle import ufp_oracle as U

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def merge_golden():
    return np.load(os.path.join(ROOT, "tests", "golden", "merge_golden.npz"))


@pytest.fixture(scope="module")
def eval_golden():
    return np.load(os.path.join(ROOT, "tests", "golden", "eval_golden.npz"))


NMS_CASES = list(range(12))


# ------------------------------------------------------------------------------------------ f.2, CPU
def test_compute_iof_equals_the_reference(merge_golden):
    g = merge_golden
    got = np.array([U.compute_iof(list(a), list(b)) for a, b in zip(g["iof/a"], g["iof/b"])], np.float64)
    assert np.array_equal(got, g["iof/value"])
    assert (g["iof/value"][:40] == 1.0).all() and (g["iof/value"][40:80] == 0.0).all()
    got32 = np.array([U.compute_iof(list(a), list(b)) for a, b in zip(g["iof/a"].astype(np.float32), g["iof/b"])], np.float64)
    assert np.array_equal(got32, g["iof/value_f32_first"])


@pytest.mark.parametrize("case", NMS_CASES)
def test_py_cpu_nms_equals_the_reference(merge_golden, case):
    g = merge_golden
    keep = U.py_cpu_nms(g["nms/%d/dets" % case], float(g["nms/%d/thr" % case]))
    assert list(keep) == list(g["nms/%d/keep" % case])


# ------------------------------------------------------------------------------------------ f.4, CPU
def _eval_cases(eg):
    return json.loads(bytes(eg["eval/meta"]).decode())


def _check_eval(eg, ci, imgs, ev, stats):
    pre = "eval/%d/" % ci
    none = eg[pre + "none"].astype(bool)
    assert len(imgs) == len(none)
    for i, e in enumerate(imgs):
        assert (e is None) == bool(none[i]), i
        if e is None:
            continue
        assert np.array_equal(np.asarray(e["dtMatches"], np.float64), eg[pre + "img%d/dtm" % i]), ("dtMatches", i)
        assert np.array_equal(np.asarray(e["gtMatches"], np.float64), eg[pre + "img%d/gtm" % i]), ("gtMatches", i)
        assert np.array_equal(np.asarray(e["dtIgnore"]).astype(np.uint8), eg[pre + "img%d/dtig" % i]), ("dtIgnore", i)
        assert np.array_equal(np.asarray(e["gtIgnore"]).astype(np.uint8), eg[pre + "img%d/gtig" % i]), ("gtIgnore", i)
        assert list(e["dtIds"]) + [-1] + list(e["gtIds"]) == list(eg[pre + "img%d/ids" % i]), ("ids", i)
    for k in ("precision", "recall", "scores"):
        assert np.array_equal(np.asarray(ev[k], np.float64), eg[pre + k]), k
    assert np.array_equal(np.asarray(stats, np.float64), eg[pre + "stats"])


def _case_inputs(meta):
    from tests.test_cocoeval import random_case
    ds, res = random_case(meta["seed"], **meta["kw"])
    area = CO.DRONE_AREA if meta["area"] == "drone" else CO.COCO_AREA
    return ds, res, tuple(meta["max_dets"]), area, meta["use_cats"]


@pytest.mark.parametrize("ci", range(7))
def test_cocoeval_oracle_equals_the_reference(eval_golden, ci):
    meta = _eval_cases(eval_golden)[ci]
    ds, res, max_dets, area, use_cats = _case_inputs(meta)
    stats, ev, imgs, pe = CO.coco_eval(copy.deepcopy(ds), res, max_dets=max_dets, area_rng=area, use_cats=use_cats)
    assert len(imgs) == meta["n_eval"]
    _check_eval(eval_golden, ci, imgs, ev, stats)


# ------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("ci", range(7))
def test_hip_cocoeval_equals_the_reference(eval_golden, ci):
    from glsdet_amd.eval import COCO, COCOeval
    meta = _eval_cases(eval_golden)[ci]
    ds, res, max_dets, area, use_cats = _case_inputs(meta)
    gt = COCO(copy.deepcopy(ds))
    E = COCOeval(gt, gt.loadRes(res), "bbox")
    E.params.maxDets = list(max_dets)
    E.params.areaRng = [list(r) for r in area]
    E.params.useCats = use_cats
    E.evaluate()
    E.accumulate()
    E.summarize()
    _check_eval(eval_golden, ci, E.evalImgs, E.eval, E.stats)


@pytest.mark.gpu
@pytest.mark.parametrize("case", NMS_CASES)
def test_hip_merge_nms_equals_the_reference_py_cpu_nms(merge_golden, case):
    """glsdet_ufp_backmap_merge with ONE chip that covers the whole canvas at magnification 1 and offset 0 is the
    identity back-mapping followed by the reference's py_cpu_nms.  The device works in float32: float32 fixture
    inputs are compared with the fixture's keep list directly, float64 ones are rounded to float32 first and
    compared with the oracle (itself pinned to the fixture above) on the rounded rows.  Inputs whose scores tie
    are compared as sets of score sequences only: the reference's order among equal scores is an artefact of
    numpy's unstable default sort."""
    import torch
    from glsdet_amd.ufp.stage2 import UfpSecondStage
    g = merge_golden
    dets = np.asarray(g["nms/%d/dets" % case])
    thr = float(g["nms/%d/thr" % case])
    d32 = dets.astype(np.float32)
    # the back-mapping rebuilds the far corner as x + (x2 - x1) (ufpmp_det_eval.py:291-295), in float32 on the device
    d32[:, 2] = d32[:, 0] + (d32[:, 2] - d32[:, 0])
    d32[:, 3] = d32[:, 1] + (d32[:, 3] - d32[:, 1])
    assert d32[:, :2].min() > 0
    keep = U.py_cpu_nms(d32, thr)             # the oracle is pinned to the reference's keep lists by the CPU test above
    want = d32[keep]
    max_det = len(d32)
    rows = np.zeros((max_det, 7), np.float32)
    rows[:, :5] = dets.astype(np.float32)
    rows[:, 5] = d32[:, 4]
    count = torch.tensor([max_det, max_det], dtype=torch.int32).cuda()
    chip = [[0, 0, 4096, 4096, 0, 0, 1]]
    out = UfpSecondStage().merge(torch.from_numpy(rows).cuda(), count, chip, 1, nms_thr=thr)[0]
    if len(np.unique(d32[:, 4])) == len(d32):
        assert len(out) == len(want)
        np.testing.assert_array_equal(out.astype(np.float32), want[:, :5])
    else:
        assert np.all(np.diff(out[:, 4]) <= 0)                 # still in descending score order
