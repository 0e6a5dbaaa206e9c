"""result formats and the bbox COCO evaluation protocol.

pycocotools is not installed and the reference's vendored copy cannot be imported (it needs
pycocotools._mask), so the oracle is PARITY UNPINNED: it is pinned here by hand-derived known answers;
the GPU path is then held bit-exact to the oracle on random data sets."""
import os

import numpy as np
import pytest

from oracle import cocoeval_oracle as CO
from oracle import glsdet_oracle as O


# ---------------------------------------------------------------------------------------- data
def _dataset(images, cats, gts):
    """gts: (image_id, category_id, [x,y,w,h], iscrowd) -> COCO dict; annotation ids from `first_id`."""
    anns = [dict(id=i + 1, image_id=im, category_id=c, bbox=list(bb), area=bb[2] * bb[3], iscrowd=cr)
            for i, (im, c, bb, cr) in enumerate(gts)]
    return dict(images=[dict(id=i) for i in images], categories=[dict(id=c) for c in cats], annotations=anns)


def _res(image_id, cat, bb, score):
    return dict(image_id=image_id, category_id=cat, bbox=list(bb), score=score)


def random_case(seed, n_img=6, n_cat=3, max_gt=9, max_dt=14, crowd=0.15, zero_id=False, ties=True, big=False):
    r = np.random.default_rng(seed)
    gts, res = [], []
    for im in range(n_img):
        if r.random() < 0.15:
            continue                                      # an image without any ground truth
        for c in range(n_cat):
            ng = int(r.integers(0, max_gt + 1))
            boxes = []
            for _ in range(ng):
                w, h = (r.integers(4, 160, 2) if big else r.integers(4, 60, 2)).tolist()
                x, y = r.integers(0, 200, 2).tolist()
                boxes.append([float(x), float(y), float(w), float(h)])
                gts.append((im, c, boxes[-1], int(r.random() < crowd)))
            nd = int(r.integers(0, max_dt + 1))
            for _ in range(nd):
                if boxes and r.random() < 0.7:
                    b = list(boxes[int(r.integers(0, len(boxes)))])
                    if r.random() < 0.6:                  # jitter: half-integer steps keep exact IoU ties likely
                        b = [b[0] + r.integers(-6, 7) * 0.5, b[1] + r.integers(-6, 7) * 0.5,
                             max(1.0, b[2] + r.integers(-6, 7) * 0.5), max(1.0, b[3] + r.integers(-6, 7) * 0.5)]
                else:
                    b = [float(v) for v in r.integers(0, 200, 2)] + [float(v) for v in r.integers(2, 80, 2)]
                s = float(np.round(r.random(), 1)) if (ties and r.random() < 0.3) else float(r.random())
                res.append(_res(im, c, b, s))
    ds = _dataset(range(n_img), range(n_cat), gts)
    if zero_id and ds["annotations"]:
        ds["annotations"][0]["id"] = 0
    return ds, res


# ---------------------------------------------------------------------------------------- oracle: known answers
def test_bb_iou_known_values():
    d = [[0, 0, 10, 10], [5, 0, 10, 10], [20, 20, 4, 4]]
    g = [[0, 0, 10, 10], [0, 0, 20, 20]]
    iou = CO.bb_iou(d, g, [0, 0])
    assert iou[0, 0] == 1.0 and iou[1, 0] == 50 / 150 and iou[2, 0] == 0.0
    assert iou[0, 1] == 100 / 400 and iou[1, 1] == 100 / 400
    crowd = CO.bb_iou(d, g, [0, 1])                       # union := detection area for a crowd ground truth
    assert crowd[0, 1] == 1.0 and crowd[1, 1] == 1.0 and crowd[2, 1] == 0.0
    assert CO.bb_iou([[0, 0, 10, 10]], [[10, 0, 5, 5]], [0])[0, 0] == 0.0       # touching edges do not overlap


def test_perfect_detections_score_one():
    ds = _dataset([0, 1], [0, 1], [(0, 0, [0, 0, 20, 20], 0), (0, 1, [50, 50, 40, 40], 0), (1, 0, [5, 5, 100, 100], 0)])
    res = [_res(a["image_id"], a["category_id"], a["bbox"], 0.9) for a in ds["annotations"]]
    stats, ev, _, _ = CO.coco_eval(ds, res)
    # precision is tp / (tp + fp + eps): one ulp below 1 (the reference's np.spacing(1) guard)
    assert np.allclose(stats[[0, 1, 2, 3, 4, 5]], 1.0, rtol=0, atol=1e-15)      # 400, 1600, 10000 px: one gt per area range
    assert stats[8] == 1.0 and stats[6] == 1.0                                  # one gt per (image, category): AR@1 too


def test_hand_derived_precision_recall():
    # one image, one category, two large ground truths; detections: hit (.9), miss (.8), hit (.7)
    ds = _dataset([0], [0], [(0, 0, [0, 0, 100, 100], 0), (0, 0, [200, 0, 100, 100], 0)])
    res = [_res(0, 0, [0, 0, 100, 100], 0.9), _res(0, 0, [500, 500, 100, 100], 0.8), _res(0, 0, [200, 0, 100, 100], 0.7)]
    stats, ev, imgs, p = CO.coco_eval(ds, res)
    # tp = 1,1,2  fp = 0,1,1  recall = .5,.5,1  precision envelope = 1, 2/3, 2/3
    ap = (51 * 1.0 + 50 * (2.0 / 3.0)) / 101
    assert abs(stats[0] - ap) < 1e-12 and abs(stats[1] - ap) < 1e-12 and abs(stats[2] - ap) < 1e-12
    assert stats[3] == -1 and stats[4] == -1 and abs(stats[5] - ap) < 1e-12     # no small / medium ground truth
    assert stats[6] == 0.5 and stats[7] == 1.0 and stats[8] == 1.0              # AR@1 sees only the best detection
    pr = ev["precision"][0, :, 0, 0, 2]
    assert np.allclose(pr[:51], 1.0, rtol=0, atol=1e-15) and np.allclose(pr[51:], 2 / 3, rtol=0, atol=1e-15)
    assert list(ev["scores"][0, [0, 50, 51, 100], 0, 0, 2]) == [0.9, 0.9, 0.7, 0.7]


def test_iou_threshold_sweep():
    # IoU = 62/100... a 100x100 box shifted by 23.5 px: inter 76.5*100, union 2*10000-7650 -> 0.6194
    ds = _dataset([0], [0], [(0, 0, [0, 0, 100, 100], 0)])
    stats, ev, _, _ = CO.coco_eval(ds, [_res(0, 0, [23.5, 0, 100, 100], 0.5)])
    iou = 7650 / 12350
    assert 0.6 < iou < 0.65
    assert abs(stats[0] - 0.3) < 1e-12 and abs(stats[1] - 1.0) < 1e-15 and stats[2] == 0.0          # thresholds .5 .55 .6 of ten
    assert abs(stats[8] - 0.3) < 1e-12


def test_crowd_and_ignore_rules():
    # a crowd region swallows any number of detections (neither tp nor fp); a regular gt is matched first
    ds = _dataset([0], [0], [(0, 0, [0, 0, 100, 100], 1), (0, 0, [10, 10, 30, 30], 0)])
    res = [_res(0, 0, [10, 10, 30, 30], 0.9), _res(0, 0, [50, 50, 20, 20], 0.8), _res(0, 0, [60, 60, 20, 20], 0.7),
           _res(0, 0, [300, 300, 20, 20], 0.6)]
    stats, ev, imgs, p = CO.coco_eval(ds, res)
    e = imgs[0]                                                               # category 0, area 'all', image 0
    assert list(e["gtIgnore"]) == [0, 1]                                      # the crowd sorts last
    assert list(e["dtMatches"][0]) == [2.0, 1.0, 1.0, 0.0]                    # ids: regular gt = 2, crowd = 1
    assert list(e["dtIgnore"][0]) == [False, True, True, False]
    # tp, (ignored, ignored), fp -> recall 1 at precision 1
    assert abs(stats[0] - 1.0) < 1e-15 and stats[8] == 1.0


def test_detection_outside_area_range_is_ignored_when_unmatched():
    ds = _dataset([0], [0], [(0, 0, [0, 0, 20, 20], 0)])                      # small gt (400 px)
    res = [_res(0, 0, [0, 0, 20, 20], 0.9), _res(0, 0, [100, 100, 200, 200], 0.8)]   # + a large false positive
    stats, ev, imgs, p = CO.coco_eval(ds, res)
    small = imgs[1]                                                           # area range 'small'
    assert list(small["dtIgnore"][0]) == [False, True]
    assert abs(stats[3] - 1.0) < 1e-15 and abs(stats[0] - 1.0) < 1e-15        # the false positive comes after full recall
    assert stats[5] == -1                                                     # no large gt at all


def test_max_dets_cut():
    ds = _dataset([0], [0], [(0, 0, [i * 30, 0, 20, 20], 0) for i in range(5)])
    res = [_res(0, 0, [i * 30, 0, 20, 20], 0.9 - 0.1 * i) for i in range(5)]
    stats, *_ = CO.coco_eval(ds, res, max_dets=(1, 3, 5))
    assert np.allclose(stats[[6, 7, 8]], [0.2, 0.6, 1.0], rtol=0, atol=1e-15)     # recall = tp / 5, mean over ten thresholds
    assert stats[0] == -1                                                     # _summarize(1) asks for maxDets=100: absent


# ---------------------------------------------------------------------------------------- host code on CPU
def test_accumulate_matches_oracle_on_cpu():
    """COCOeval.accumulate / summarize are host code: fed with the oracle's evalImgs they must give the
    oracle's precision / recall / scores bit for bit."""
    from glsdet_amd.eval.cocoeval import COCOeval, Params
    for seed in range(4):
        ds, res = random_case(seed, zero_id=(seed == 2))
        stats, ev, imgs, pe = CO.coco_eval(ds, res, max_dets=(1, 5, 12))
        E = COCOeval()
        for k in ("imgIds", "catIds", "maxDets", "areaRng", "useCats"):
            setattr(E.params, k, pe[k])
        import copy
        E.evalImgs, E._paramsEval = imgs, copy.deepcopy(E.params)
        E.accumulate()
        for k in ("precision", "recall", "scores"):
            assert np.array_equal(E.eval[k], ev[k]), (seed, k)
        E.summarize()
        assert np.array_equal(E.stats, stats)


def test_coco_container_and_load_res():
    from glsdet_amd.eval import COCO
    ds, res = random_case(3)
    gt = COCO(ds)
    assert gt.getImgIds() == [im["id"] for im in ds["images"]] and gt.getCatIds() == [0, 1, 2]
    ids = gt.getAnnIds(imgIds=[2, 0], catIds=[1])
    assert ids == [a["id"] for i in (2, 0) for a in ds["annotations"] if a["image_id"] == i and a["category_id"] == 1]
    dt = gt.loadRes(res)
    ref = CO.load_res(ds, res)
    assert [a["id"] for a in dt.dataset["annotations"]] == list(range(1, len(res) + 1))
    assert [a["area"] for a in dt.dataset["annotations"]] == [a["area"] for a in ref["annotations"]]
    arr = np.array([[r["image_id"]] + r["bbox"] + [r["score"], r["category_id"]] for r in res])
    dt2 = gt.loadRes(arr)
    assert [a["bbox"] for a in dt2.dataset["annotations"]] == [a["bbox"] for a in dt.dataset["annotations"]]
    with pytest.raises(AssertionError):
        gt.loadRes([_res(999, 0, [0, 0, 1, 1], 0.5)])


def test_result_line_formats(tmp_path):
    from glsdet_amd.eval import coco_records, detection_line, parse_detection_results, parse_per_class, write_detection_results
    assert detection_line("car", np.float32(0.87654321), 10.9, 20.1, 30.99, 40.5) == "car 0.8765 10 20 30 40\n"
    assert detection_line("van", np.float32(1.2345678e-05), -0.5, 3, 4, 5) == "van 1.2345 0 3 4 5\n"     # the reference's lossy cut
    rows = np.array([[20.1, 10.9, 40.5, 30.99, 0.9, 0.8, 3], [1, 2, 3, 4, 0.5, 0.5, 0]], np.float32)   # top,left,bottom,right
    names = ["pedestrian", "people", "bicycle", "car"]
    p = str(tmp_path / "a.txt")
    assert write_detection_results(p, rows, names) == 2
    text = open(p).read().splitlines()
    assert text[0] == "car %s 10 20 30 40" % str(np.float32(0.9) * np.float32(0.8))[:6] and text[1] == "pedestrian 0.25 2 1 4 3"
    index = {n: i for i, n in enumerate(names)}
    parsed = parse_detection_results(p, index)
    assert parsed[0] == [10.0, 20.0, 30.0, 40.0, float(text[0].split()[1]), 3] and parsed[1][-1] == 0
    per = parse_per_class(p, index, min_score=0.3)
    assert per[3] == [parsed[0][:5]] and per[0] == []
    assert write_detection_results(str(tmp_path / "b.txt"), None, names) == 0 and open(str(tmp_path / "b.txt")).read() == ""
    rec = coco_records(7, [np.array([[1.9, 2.9, 11.2, 12.7, 0.5]]), np.zeros((0, 5)), np.array([[-0.5, 0.2, 4.9, 5.1, 0.25]])])
    assert rec == [dict(image_id=7, category_id=0, score=0.5, bbox=[1, 2, 10, 10]),
                   dict(image_id=7, category_id=2, score=0.25, bbox=[0, 0, 4, 5])]


# ---------------------------------------------------------------------------------------- GPU parity
def _run_gpu(ds, res, max_dets, area, use_cats):
    from glsdet_amd.eval import COCO, COCOeval
    gt = COCO(ds)
    E = COCOeval(gt, gt.loadRes(res), "bbox")
    E.params.maxDets = list(max_dets)
    E.params.areaRng = [list(a) for a in area]
    E.params.useCats = use_cats
    E.evaluate()
    E.accumulate()
    E.summarize()
    return E


@pytest.mark.gpu
@pytest.mark.parametrize("seed,kw", [(0, {}), (1, dict(zero_id=True)), (2, dict(crowd=0.5)), (3, dict(n_img=1, n_cat=1, max_gt=70, max_dt=150)),
                                     (4, dict(max_gt=2, max_dt=40)), (5, dict(big=True)), (6, dict(n_img=12, n_cat=10)), (7, dict(crowd=0.0, ties=False))])
@pytest.mark.parametrize("use_cats", [1, 0])
def test_evaluate_bit_exact_vs_oracle(seed, kw, use_cats):
    import copy
    ds, res = random_case(100 + seed, **kw)
    max_dets = (1, 10, 25) if seed != 3 else (10, 100, 120)
    area = CO.DRONE_AREA if seed == 5 else CO.COCO_AREA
    stats, ev, imgs, pe = CO.coco_eval(copy.deepcopy(ds), res, max_dets=max_dets, area_rng=area, use_cats=use_cats)
    _, _, ious = CO.evaluate(copy.deepcopy(ds), CO.load_res(ds, res), dict(pe))
    E = _run_gpu(copy.deepcopy(ds), res, max_dets, area, use_cats)
    assert len(E.evalImgs) == len(imgs)
    for got, want in zip(E.evalImgs, imgs):
        assert (got is None) == (want is None)
        if want is None:
            continue
        for k in ("image_id", "category_id", "maxDet", "dtIds", "gtIds", "dtScores"):
            assert got[k] == want[k], k
        for k in ("dtMatches", "gtMatches", "gtIgnore", "dtIgnore"):
            assert np.array_equal(np.asarray(got[k]), np.asarray(want[k])), (k, got["image_id"], got["category_id"], got["aRng"])
    for key, want in ious.items():
        got = E.ious[key]
        if len(want) == 0 or np.asarray(want).size == 0:
            assert np.asarray(got).size == 0
        else:
            assert np.array_equal(got, want), key                    # fp64, unfused: bit for bit
    for k in ("precision", "recall", "scores"):
        assert np.array_equal(E.eval[k], ev[k]), k
    assert np.array_equal(E.stats, stats)


@pytest.mark.gpu
def test_evaluate_edge_cases():
    # no detections at all; no ground truth at all; nothing at all
    ds = _dataset([0, 1], [0], [(0, 0, [0, 0, 20, 20], 0)])
    E = _run_gpu(ds, [], (1, 10, 100), CO.COCO_AREA, 1)
    assert E.stats[0] == 0.0 and E.stats[8] == 0.0
    ds2 = _dataset([0, 1], [0], [])
    E = _run_gpu(ds2, [_res(1, 0, [0, 0, 5, 5], 0.5)], (1, 10, 100), CO.COCO_AREA, 1)
    assert np.all(E.stats == -1)
    want = CO.coco_eval(ds2, [_res(1, 0, [0, 0, 5, 5], 0.5)])[0]
    assert np.array_equal(E.stats, want)
    E = _run_gpu(_dataset([0], [0], []), [], (1, 10, 100), CO.COCO_AREA, 1)
    assert np.all(E.stats == -1) and all(e is None for e in E.evalImgs)


@pytest.mark.gpu
def test_known_answer_on_gpu():
    ds = _dataset([0], [0], [(0, 0, [0, 0, 100, 100], 0), (0, 0, [200, 0, 100, 100], 0)])
    res = [_res(0, 0, [0, 0, 100, 100], 0.9), _res(0, 0, [500, 500, 100, 100], 0.8), _res(0, 0, [200, 0, 100, 100], 0.7)]
    E = _run_gpu(ds, res, (1, 10, 100), CO.COCO_AREA, 1)
    assert abs(E.stats[0] - (51 + 50 * 2 / 3) / 101) < 1e-12 and E.stats[6] == 0.5 and E.stats[7] == 1.0


@pytest.mark.gpu
def test_visdrone_sized_evaluation_agrees_with_oracle_summary():
    """A denser set (one category of the oracle's cost would take minutes in full): 40 images x 10
    categories, up to 500 detections per image in total, maxDets of the two-stage eval."""
    ds, res = random_case(77, n_img=40, n_cat=10, max_gt=25, max_dt=50)
    stats, ev, imgs, pe = CO.coco_eval(ds, res, max_dets=(10, 100, 500))
    E = _run_gpu(ds, res, (10, 100, 500), CO.COCO_AREA, 1)
    assert np.array_equal(E.stats, stats)
    assert np.array_equal(E.eval["precision"], ev["precision"])


@pytest.mark.gpu
def test_result_merger_matches_batched_nms(tmp_path):
    from glsdet_amd.eval import ResultMerger, VISDRONE_CLASSES, detection_line
    r = np.random.default_rng(5)
    m = ResultMerger(capacity=2048)
    for trial in range(3):
        n = [0, 300, 1500][trial]
        xy = r.uniform(0, 600, (n, 2)).astype(np.float32)
        wh = r.uniform(8, 90, (n, 2)).astype(np.float32)
        rows = np.concatenate([xy, xy + wh, r.uniform(0.05, 1, (n, 1)).astype(np.float32),
                               r.integers(0, 10, (n, 1)).astype(np.float32)], axis=1)
        kept = m.merge_rows(rows)
        want = O.batched_nms(rows[:, :4], rows[:, 4], rows[:, 5].astype(np.int64), 0.65)
        assert np.array_equal(kept, rows[want])
    # a score of exactly 0 (the '0.0000' cut of the reference's 6-character score string) survives like any other row
    rows = np.array([[10, 10, 50, 50, 0.0, 3], [200, 200, 260, 250, 0.0, 0], [11, 11, 51, 51, 0.4, 3]], np.float32)
    kept = m.merge_rows(rows)
    assert np.array_equal(kept, rows[[2, 1]])
    # file level: two result directories -> merged directory
    d1, d2, out = tmp_path / "a", tmp_path / "b", tmp_path / "o"
    d1.mkdir(); d2.mkdir()
    lines1 = [detection_line("car", 0.9, 10, 10, 50, 50), detection_line("bus", 0.8, 100, 100, 180, 160)]
    lines2 = [detection_line("car", 0.7, 12, 11, 52, 49), detection_line("people", 0.6, 10, 10, 50, 50)]
    (d1 / "img.txt").write_text("".join(lines1)); (d2 / "img.txt").write_text("".join(lines2))
    assert m.merge_dirs([str(d1), str(d2)], str(out)) == 3
    # merge_results.py:166 prints float(<float32 tensor element>): the float32 value at double precision
    f32 = lambda v: float(np.float32(v))
    assert (out / "img.txt").read_text().splitlines() == ["car %s 10 10 50 50" % f32(0.9), "bus %s 100 100 180 160" % f32(0.8),
                                                           "people %s 10 10 50 50" % f32(0.6)]
