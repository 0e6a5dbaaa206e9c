"""CPU oracle of the bbox COCO evaluation protocol  --  TEST INFRASTRUCTURE ONLY.

Restates, in plain Python loops over dicts (slow, small cases only), what ufp/ufpmp_det_eval.py:328-338
runs through pycocotools (`COCO.loadRes`, `COCOeval(gt, dt, 'bbox')`, `params.maxDets = [10, 100, 500]`,
evaluate / accumulate / summarize).  The algorithm text the reference vendors is
drone/models/core/cocoeval.py (pycocotools 2.0's cocoeval.py with its area ranges edited); the functions
below follow it:

    prepare        :84-119    ground truths / detections grouped per (image, category), ignore flags
    compute_iou    :163-190   detections by descending score (mergesort), cut to the largest maxDets
    bb_iou                    pycocotools common/maskApi.c `bbIou` (NOT under /root/reference): xywh boxes,
                              intersection over union, union = detection area for a crowd ground truth
    evaluate_img   :235-313   greedy matching per IoU threshold, ignored ground truths last
    accumulate     :315-420   precision at 101 recall thresholds per (T, K, A, M), recall, scores
    summarize      :422-470   the 12 numbers

PINNED (round 2) for everything but the IoU arithmetic: tests/golden/eval_golden.npz holds evalImgs, precision /
recall / scores and the 12 stats produced by the reference's OWN vendored COCOeval (`_prepare`, `computeIoU`'s
score ordering and maxDets cut, `evaluateImg`, `accumulate`, `summarize`) on seeded data sets
(tests/golden/make_golden.py::eval_cases; tests/test_pinned_goldens.py holds this file to them exactly).
STILL UNPINNED: `bb_iou`.  The reference reaches it through pycocotools' compiled `_mask.iou`, which is not
installed; the fixture generator routes the vendored module's `maskUtils.iou` to this restatement of
maskApi.c's bbIou, so the fixtures say nothing about it (hand-derived values: tests/test_cocoeval.py).  `COCO_AREA` are pycocotools' own ranges (what the two-stage eval
runs with); `DRONE_AREA` are the edited ones of the vendored copy (:507-508)."""
from __future__ import annotations

import copy
from collections import defaultdict
from typing import Dict, List, Optional, Sequence

import numpy as np

COCO_AREA = [[0 ** 2, 1e5 ** 2], [0 ** 2, 32 ** 2], [32 ** 2, 96 ** 2], [96 ** 2, 1e5 ** 2]]
DRONE_AREA = [[0 ** 2, 1e5 ** 2], [0 ** 2, 32 ** 2], [32 ** 2, 1e5 ** 2], [1e5 ** 2, 1e5 ** 2]]      # cocoeval.py:508
AREA_LABELS = ["all", "small", "medium", "large"]


def default_params(max_dets: Sequence[int] = (1, 10, 100), area_rng=None) -> dict:
    """Params.setDetParams (cocoeval.py:502-512)."""
    return dict(imgIds=[], catIds=[],
                iouThrs=np.linspace(.5, 0.95, int(np.round((0.95 - .5) / .05)) + 1, endpoint=True),
                recThrs=np.linspace(.0, 1.00, int(np.round((1.00 - .0) / .01)) + 1, endpoint=True),
                maxDets=list(max_dets), areaRng=[list(r) for r in (area_rng or COCO_AREA)],
                areaRngLbl=list(AREA_LABELS), useCats=1)


def load_res(gt_dataset: dict, results: List[dict]) -> dict:
    """COCO.loadRes for bbox results (pycocotools coco.py): ids from 1, area = w*h, iscrowd = 0."""
    img_ids = {im["id"] for im in gt_dataset["images"]}
    assert {r["image_id"] for r in results} <= img_ids, "Results do not correspond to current coco set"
    anns = copy.deepcopy(results)
    for i, ann in enumerate(anns):
        bb = ann["bbox"]
        ann["area"] = bb[2] * bb[3]
        ann["id"] = i + 1
        ann["iscrowd"] = 0
    return dict(images=list(gt_dataset["images"]), categories=copy.deepcopy(gt_dataset["categories"]), annotations=anns)


def bb_iou(dts: Sequence[Sequence[float]], gts: Sequence[Sequence[float]], iscrowd: Sequence[int]) -> np.ndarray:
    out = np.zeros((len(dts), len(gts)))
    for g, G in enumerate(gts):
        ga = G[2] * G[3]
        for d, D in enumerate(dts):
            da = D[2] * D[3]
            w = min(D[2] + D[0], G[2] + G[0]) - max(D[0], G[0])
            if w <= 0:
                continue
            h = min(D[3] + D[1], G[3] + G[1]) - max(D[1], G[1])
            if h <= 0:
                continue
            i = w * h
            u = da if iscrowd[g] else da + ga - i
            out[d, g] = i / u
    return out


def _anns_of(dataset: dict, img_ids, cat_ids, use_cats) -> List[dict]:
    """loadAnns(getAnnIds(imgIds, catIds)): image by image in the order of img_ids, annotation order inside."""
    by_img = defaultdict(list)
    for ann in dataset["annotations"]:
        by_img[ann["image_id"]].append(ann)
    out = [a for i in img_ids if i in by_img for a in by_img[i]]
    if use_cats and len(cat_ids) > 0:
        out = [a for a in out if a["category_id"] in cat_ids]
    return out


def prepare(gt_dataset: dict, dt_dataset: dict, p: dict):
    gts = [dict(a) for a in _anns_of(gt_dataset, p["imgIds"], p["catIds"], p["useCats"])]
    dts = [dict(a) for a in _anns_of(dt_dataset, p["imgIds"], p["catIds"], p["useCats"])]
    for gt in gts:
        gt["ignore"] = bool("iscrowd" in gt and gt["iscrowd"])
    G, D = defaultdict(list), defaultdict(list)
    for gt in gts:
        G[gt["image_id"], gt["category_id"]].append(gt)
    for dt in dts:
        D[dt["image_id"], dt["category_id"]].append(dt)
    return G, D


def _lists(G, D, p, img, cat):
    if p["useCats"]:
        return G[img, cat], D[img, cat]
    return [a for c in p["catIds"] for a in G[img, c]], [a for c in p["catIds"] for a in D[img, c]]


def compute_iou(G, D, p, img, cat):
    gt, dt = _lists(G, D, p, img, cat)
    if len(gt) == 0 and len(dt) == 0:
        return []
    order = np.argsort([-d["score"] for d in dt], kind="mergesort")
    dt = [dt[i] for i in order][:p["maxDets"][-1]]
    return bb_iou([d["bbox"] for d in dt], [g["bbox"] for g in gt], [int(g.get("iscrowd", 0)) for g in gt])


def evaluate_img(G, D, p, ious, img, cat, a_rng, max_det) -> Optional[dict]:
    gt, dt = _lists(G, D, p, img, cat)
    if len(gt) == 0 and len(dt) == 0:
        return None
    ig = [1 if (g["ignore"] or g["area"] < a_rng[0] or g["area"] > a_rng[1]) else 0 for g in gt]
    gtind = np.argsort(ig, kind="mergesort")
    gt = [gt[i] for i in gtind]
    dtind = np.argsort([-d["score"] for d in dt], kind="mergesort")
    dt = [dt[i] for i in dtind[0:max_det]]
    iscrowd = [int(g.get("iscrowd", 0)) for g in gt]
    iou = ious[img, cat][:, gtind] if len(ious[img, cat]) > 0 else ious[img, cat]
    T, Gn, Dn = len(p["iouThrs"]), len(gt), len(dt)
    gtm, dtm = np.zeros((T, Gn)), np.zeros((T, Dn))
    gt_ig = np.array([ig[i] for i in gtind])
    dt_ig = np.zeros((T, Dn))
    if not len(iou) == 0:
        for ti, t in enumerate(p["iouThrs"]):
            for di, d in enumerate(dt):
                best = min([t, 1 - 1e-10])
                m = -1
                for gi in range(Gn):
                    if gtm[ti, gi] > 0 and not iscrowd[gi]:
                        continue
                    if m > -1 and gt_ig[m] == 0 and gt_ig[gi] == 1:
                        break
                    if iou[di, gi] < best:
                        continue
                    best = iou[di, gi]
                    m = gi
                if m == -1:
                    continue
                dt_ig[ti, di] = gt_ig[m]
                dtm[ti, di] = gt[m]["id"]
                gtm[ti, m] = d["id"]
    outside = np.array([d["area"] < a_rng[0] or d["area"] > a_rng[1] for d in dt]).reshape((1, len(dt)))
    dt_ig = np.logical_or(dt_ig, np.logical_and(dtm == 0, np.repeat(outside, T, 0)))
    return dict(image_id=img, category_id=cat, aRng=a_rng, maxDet=max_det, dtIds=[d["id"] for d in dt],
                gtIds=[g["id"] for g in gt], dtMatches=dtm, gtMatches=gtm, dtScores=[d["score"] for d in dt],
                gtIgnore=gt_ig, dtIgnore=dt_ig)


def evaluate(gt_dataset: dict, dt_dataset: dict, params: dict):
    """-> (evalImgs in the reference's order: category, area range, image; the params as evaluated; ious)."""
    p = dict(params)
    p["imgIds"] = list(np.unique(p["imgIds"]))
    if p["useCats"]:
        p["catIds"] = list(np.unique(p["catIds"]))
    p["maxDets"] = sorted(p["maxDets"])
    G, D = prepare(gt_dataset, dt_dataset, p)
    cats = p["catIds"] if p["useCats"] else [-1]
    ious = {(i, c): compute_iou(G, D, p, i, c) for i in p["imgIds"] for c in cats}
    md = p["maxDets"][-1]
    imgs = [evaluate_img(G, D, p, ious, i, c, a, md) for c in cats for a in p["areaRng"] for i in p["imgIds"]]
    return imgs, p, ious


def accumulate(eval_imgs: List[Optional[dict]], p: dict) -> Dict[str, np.ndarray]:
    cats = p["catIds"] if p["useCats"] else [-1]
    T, R, K, A, M = len(p["iouThrs"]), len(p["recThrs"]), len(cats), len(p["areaRng"]), len(p["maxDets"])
    precision, recall, scores = -np.ones((T, R, K, A, M)), -np.ones((T, K, A, M)), -np.ones((T, R, K, A, M))
    I = len(p["imgIds"])
    for k in range(K):
        for a in range(A):
            for m, max_det in enumerate(p["maxDets"]):
                E = [eval_imgs[k * A * I + a * I + i] for i in range(I)]
                E = [e for e in E if e is not None]
                if len(E) == 0:
                    continue
                dt_scores = np.concatenate([e["dtScores"][0:max_det] for e in E])
                inds = np.argsort(-dt_scores, kind="mergesort")
                sorted_scores = dt_scores[inds]
                dtm = np.concatenate([e["dtMatches"][:, 0:max_det] for e in E], axis=1)[:, inds]
                dt_ig = np.concatenate([e["dtIgnore"][:, 0:max_det] for e in E], axis=1)[:, inds]
                gt_ig = np.concatenate([e["gtIgnore"] for e in E])
                npig = np.count_nonzero(gt_ig == 0)
                if npig == 0:
                    continue
                tps = np.logical_and(dtm, np.logical_not(dt_ig))
                fps = np.logical_and(np.logical_not(dtm), np.logical_not(dt_ig))
                tp_sum = np.cumsum(tps, axis=1).astype(dtype=float)
                fp_sum = np.cumsum(fps, axis=1).astype(dtype=float)
                for t, (tp, fp) in enumerate(zip(tp_sum, fp_sum)):
                    nd = len(tp)
                    rc = tp / npig
                    pr = (tp / (fp + tp + np.spacing(1))).tolist()
                    q, ss = [0.0] * R, np.zeros((R,))
                    recall[t, k, a, m] = rc[-1] if nd else 0
                    for i in range(nd - 1, 0, -1):
                        if pr[i] > pr[i - 1]:
                            pr[i - 1] = pr[i]
                    for ri, pi in enumerate(np.searchsorted(rc, p["recThrs"], side="left")):
                        if pi >= nd:          # the reference's try/except IndexError: the rest stays 0
                            break
                        q[ri] = pr[pi]
                        ss[ri] = sorted_scores[pi]
                    precision[t, :, k, a, m] = np.array(q)
                    scores[t, :, k, a, m] = ss
    return dict(precision=precision, recall=recall, scores=scores, counts=[T, R, K, A, M])


def summarize(ev: Dict[str, np.ndarray], p: dict) -> np.ndarray:
    def one(ap=1, iou_thr=None, area="all", max_dets=100):
        aind = [i for i, l in enumerate(p["areaRngLbl"]) if l == area]
        mind = [i for i, m in enumerate(p["maxDets"]) if m == max_dets]
        s = ev["precision"] if ap == 1 else ev["recall"]
        if iou_thr is not None:
            s = s[np.where(iou_thr == p["iouThrs"])[0]]
        s = s[:, :, :, aind, mind] if ap == 1 else s[:, :, aind, mind]
        return -1 if len(s[s > -1]) == 0 else np.mean(s[s > -1])

    md = p["maxDets"]
    return np.array([one(1), one(1, iou_thr=.5, max_dets=md[2]), one(1, iou_thr=.75, max_dets=md[2]),
                     one(1, area="small", max_dets=md[2]), one(1, area="medium", max_dets=md[2]),
                     one(1, area="large", max_dets=md[2]), one(0, max_dets=md[0]), one(0, max_dets=md[1]),
                     one(0, max_dets=md[2]), one(0, area="small", max_dets=md[2]),
                     one(0, area="medium", max_dets=md[2]), one(0, area="large", max_dets=md[2])], dtype=np.float64)


def coco_eval(gt_dataset: dict, results: List[dict], max_dets=(1, 10, 100), area_rng=None, use_cats=1):
    """The whole protocol of ufpmp_det_eval.py:328-338 -> (stats[12], accumulated dict, evalImgs, params)."""
    p = default_params(max_dets, area_rng)
    p["imgIds"] = sorted(im["id"] for im in gt_dataset["images"])
    p["catIds"] = sorted(c["id"] for c in gt_dataset["categories"])
    p["useCats"] = use_cats
    imgs, pe, _ = evaluate(gt_dataset, load_res(gt_dataset, results), p)
    ev = accumulate(imgs, pe)
    return summarize(ev, pe), ev, imgs, pe
