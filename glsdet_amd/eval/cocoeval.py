"""bbox COCOeval with the per-(image, category) work on the GPU.

Same surface as the class the reference evaluates with (ufp/ufpmp_det_eval.py:333-338 through
pycocotools; the vendored text is drone/models/core/cocoeval.py):

    E = COCOeval(cocoGt, cocoDt, 'bbox'); E.params.maxDets = [10, 100, 500]
    E.evaluate(); E.accumulate(); E.summarize(); E.stats

evaluate() packs every (image, category) pair into flat fp64 arrays, runs IoU + greedy matching for all
pairs, area ranges and IoU thresholds in ONE `glsdet_coco_match` call and unpacks `evalImgs` in the
reference's layout.  accumulate() is the reference's arithmetic vectorised over the recall thresholds
(same fp64 operations, same results).  There is no CPU matching path: without the HIP library evaluate()
raises."""
from __future__ import annotations

import copy
import datetime
from collections import defaultdict
from typing import List, Optional

import numpy as np
import torch

from .. import _lib

COCO_AREA = [[0 ** 2, 1e5 ** 2], [0 ** 2, 32 ** 2], [32 ** 2, 96 ** 2], [96 ** 2, 1e5 ** 2]]
DRONE_AREA = [[0 ** 2, 1e5 ** 2], [0 ** 2, 32 ** 2], [32 ** 2, 1e5 ** 2], [1e5 ** 2, 1e5 ** 2]]   # cocoeval.py:508


class Params:
    """Params.setDetParams (cocoeval.py:502-512).  `areaRng` defaults to pycocotools' own ranges -- what
    the two-stage evaluation runs with; `Params.DRONE_AREA` are the ranges of the vendored copy."""
    COCO_AREA, DRONE_AREA = COCO_AREA, DRONE_AREA

    def __init__(self, iouType: str = "bbox"):
        if iouType != "bbox":
            raise ValueError("iouType %r not supported (bbox only)" % (iouType,))
        self.imgIds: list = []
        self.catIds: list = []
        self.iouThrs = np.linspace(.5, 0.95, int(np.round((0.95 - .5) / .05)) + 1, endpoint=True)
        self.recThrs = np.linspace(.0, 1.00, int(np.round((1.00 - .0) / .01)) + 1, endpoint=True)
        self.maxDets = [1, 10, 100]
        self.areaRng = [list(r) for r in COCO_AREA]
        self.areaRngLbl = ["all", "small", "medium", "large"]
        self.useCats = 1
        self.iouType = iouType
        self.useSegm = None


class COCOeval:
    def __init__(self, cocoGt=None, cocoDt=None, iouType: str = "bbox", device: str = "cuda:0"):
        if iouType != "bbox":
            raise ValueError("iouType %r not supported (bbox only)" % (iouType,))
        self.cocoGt, self.cocoDt = cocoGt, cocoDt
        self.evalImgs: list = []
        self.eval: dict = {}
        self._gts, self._dts = defaultdict(list), defaultdict(list)
        self.params = Params(iouType)
        self._paramsEval: Optional[Params] = None
        self.stats = []
        self.ious: dict = {}
        self.device = device
        if cocoGt is not None:
            self.params.imgIds = sorted(cocoGt.getImgIds())
            self.params.catIds = sorted(cocoGt.getCatIds())

    # ------------------------------------------------------------------ cocoeval.py:84-119
    def _prepare(self) -> None:
        p = self.params
        if p.useCats:
            gts = self.cocoGt.loadAnns(self.cocoGt.getAnnIds(imgIds=p.imgIds, catIds=p.catIds))
            dts = self.cocoDt.loadAnns(self.cocoDt.getAnnIds(imgIds=p.imgIds, catIds=p.catIds))
        else:
            gts = self.cocoGt.loadAnns(self.cocoGt.getAnnIds(imgIds=p.imgIds))
            dts = self.cocoDt.loadAnns(self.cocoDt.getAnnIds(imgIds=p.imgIds))
        for gt in gts:
            gt["ignore"] = "iscrowd" in gt and gt["iscrowd"]
        self._gts, self._dts = defaultdict(list), defaultdict(list)
        for gt in gts:
            self._gts[gt["image_id"], gt["category_id"]].append(gt)
        for dt in dts:
            self._dts[dt["image_id"], dt["category_id"]].append(dt)
        self.evalImgs, self.eval = [], {}

    def _pair_lists(self, img, cat):
        p = self.params
        if p.useCats:
            return self._gts[img, cat], self._dts[img, cat]
        return ([a for c in p.catIds for a in self._gts[img, c]], [a for c in p.catIds for a in self._dts[img, c]])

    # ------------------------------------------------------------------ cocoeval.py:121-161, 163-190, 235-313
    def evaluate(self) -> None:
        p = self.params
        p.imgIds = list(np.unique(p.imgIds))
        if p.useCats:
            p.catIds = list(np.unique(p.catIds))
        p.maxDets = sorted(p.maxDets)
        self._prepare()
        cats = p.catIds if p.useCats else [-1]
        max_det = p.maxDets[-1]
        A, T, I = len(p.areaRng), len(p.iouThrs), len(p.imgIds)

        # ---- pack: pairs in (category, image) order
        pairs, dt_rows, gt_rows, gt_flags = [], [], [], []
        dt_off, gt_off, iou_off = [0], [0], [0]
        for c in cats:
            for i in p.imgIds:
                gt, dt = self._pair_lists(i, c)
                if len(dt):
                    order = np.argsort([-d["score"] for d in dt], kind="mergesort")[:max_det]
                    dt = [dt[j] for j in order]
                pairs.append((gt, dt))
                for d in dt:
                    bb = d["bbox"]
                    dt_rows.append((bb[0], bb[1], bb[2], bb[3], d["area"]))
                for g in gt:
                    bb = g["bbox"]
                    gt_rows.append((bb[0], bb[1], bb[2], bb[3], g["area"]))
                    gt_flags.append((1 if int(g.get("iscrowd", 0)) else 0) | (2 if g["ignore"] else 0) | (4 if g["id"] == 0 else 0))
                dt_off.append(dt_off[-1] + len(dt))
                gt_off.append(gt_off[-1] + len(gt))
                iou_off.append(iou_off[-1] + len(dt) * len(gt))
        out = self._match(np.asarray(dt_rows, np.float64).reshape(-1, 5), np.asarray(gt_rows, np.float64).reshape(-1, 5),
                          np.asarray(gt_flags, np.uint8), np.asarray(dt_off, np.int32), np.asarray(gt_off, np.int32),
                          np.asarray(iou_off, np.int64), np.asarray(p.areaRng, np.float64).reshape(-1, 2),
                          np.asarray(p.iouThrs, np.float64))
        ious, gt_order, gt_ignore, dt_match, dt_ignore, gt_match = out

        # ---- unpack into the reference's evalImgs (category, area range, image) and self.ious
        self.ious = {}
        self.evalImgs = [None] * (len(cats) * A * I)
        for ci, c in enumerate(cats):
            for ii, i in enumerate(p.imgIds):
                pi = ci * I + ii
                gt, dt = pairs[pi]
                D, G = len(dt), len(gt)
                if D == 0 and G == 0:
                    self.ious[i, c] = []
                    continue
                d0, g0 = dt_off[pi], gt_off[pi]
                self.ious[i, c] = ious[iou_off[pi]:iou_off[pi + 1]].reshape(D, G)
                dt_ids = [d["id"] for d in dt]
                dt_scores = [d["score"] for d in dt]
                gt_ids = np.asarray([g["id"] for g in gt], np.float64)
                dt_id_arr = np.asarray(dt_ids, np.float64)
                for a in range(A):
                    order = gt_order[a, g0:g0 + G]
                    dm = dt_match[a, :, d0:d0 + D]
                    gm = gt_match[a, :, g0:g0 + G]
                    dtm = np.where(dm >= 0, gt_ids[np.maximum(dm, 0)], 0.0) if G else np.zeros((T, D))
                    gtm = np.where(gm >= 0, dt_id_arr[np.maximum(gm, 0)], 0.0) if D else np.zeros((T, G))
                    self.evalImgs[ci * A * I + a * I + ii] = {
                        "image_id": i, "category_id": c, "aRng": p.areaRng[a], "maxDet": max_det,
                        "dtIds": dt_ids, "gtIds": [gt[j]["id"] for j in order],
                        "dtMatches": dtm, "gtMatches": gtm, "dtScores": dt_scores,
                        "gtIgnore": gt_ignore[a, g0:g0 + G].astype(np.int64), "dtIgnore": dt_ignore[a, :, d0:d0 + D].astype(bool)}
        self._paramsEval = copy.deepcopy(self.params)

    def _match(self, dt, gt, gt_flags, dt_off, gt_off, iou_off, area, thr):
        """-> host arrays (ious, gt_order[A][NG], gt_ignore[A][NG], dt_match[A][T][ND], dt_ignore, gt_match)."""
        lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.GlsdetLibraryError("glsdet_amd needs an MI355X visible to PyTorch-ROCm (no CPU fallback)")
        dev = torch.device(self.device)
        P, ND, NG, A, T = len(dt_off) - 1, len(dt), len(gt), len(area), len(thr)
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        d_box, d_area = up(dt[:, :4]), up(dt[:, 4])
        g_box, g_area, g_fl = up(gt[:, :4]), up(gt[:, 4]), up(gt_flags)
        d_off, g_off, i_off, d_rng, d_thr = up(dt_off), up(gt_off), up(iou_off), up(area), up(thr)
        ious = torch.empty(max(int(iou_off[-1]), 1), dtype=torch.float64, device=dev)
        gt_order = torch.empty(A, max(NG, 1), dtype=torch.int32, device=dev)
        gt_ignore = torch.empty(A, max(NG, 1), dtype=torch.uint8, device=dev)
        n_regular = torch.empty(A, max(P, 1), dtype=torch.int32, device=dev)
        dt_match = torch.empty(A, T, max(ND, 1), dtype=torch.int32, device=dev)
        dt_ignore = torch.empty(A, T, max(ND, 1), dtype=torch.uint8, device=dev)
        gt_match = torch.empty(A, T, max(NG, 1), dtype=torch.int32, device=dev)
        if NG == 0:
            gt_order, gt_ignore, gt_match = gt_order[:, :0], gt_ignore[:, :0], gt_match[:, :, :0]
        if ND == 0:
            dt_match, dt_ignore = dt_match[:, :, :0], dt_ignore[:, :, :0]
        ptr = lambda t: t.data_ptr() if t.numel() else None
        _lib.check(lib.glsdet_coco_match(ptr(d_box), ptr(d_area), d_off.data_ptr(), ptr(g_box), ptr(g_area), ptr(g_fl),
                                         g_off.data_ptr(), i_off.data_ptr(), P, ND, NG, d_rng.data_ptr(), A, d_thr.data_ptr(), T,
                                         ious.data_ptr(), ptr(gt_order), ptr(gt_ignore), n_regular.data_ptr(), ptr(dt_match),
                                         ptr(dt_ignore), ptr(gt_match), torch.cuda.current_stream(dev).cuda_stream), "coco_match")
        torch.cuda.current_stream(dev).synchronize()
        return (ious.cpu().numpy(), gt_order.cpu().numpy(), gt_ignore.cpu().numpy(), dt_match.cpu().numpy(),
                dt_ignore.cpu().numpy(), gt_match.cpu().numpy())

    # ------------------------------------------------------------------ cocoeval.py:315-420
    def accumulate(self, p: Optional[Params] = None) -> None:
        if not self.evalImgs:
            print("Please run evaluate() first")
        if p is None:
            p = self.params
        p.catIds = p.catIds if p.useCats == 1 else [-1]
        T, R, K, A, M = len(p.iouThrs), len(p.recThrs), len(p.catIds) if p.useCats else 1, len(p.areaRng), len(p.maxDets)
        precision, recall, scores = -np.ones((T, R, K, A, M)), -np.ones((T, K, A, M)), -np.ones((T, R, K, A, M))
        pe = self._paramsEval
        cats = pe.catIds if pe.useCats else [-1]
        set_k, set_a, set_m, set_i = set(cats), set(map(tuple, pe.areaRng)), set(pe.maxDets), set(pe.imgIds)
        k_list = [n for n, k in enumerate(p.catIds) if k in set_k]
        m_list = [m for m in p.maxDets if m in set_m]
        a_list = [n for n, a in enumerate(map(tuple, p.areaRng)) if a in set_a]
        i_list = [n for n, i in enumerate(p.imgIds) if i in set_i]
        I0, A0 = len(pe.imgIds), len(pe.areaRng)
        rec_thrs = np.asarray(p.recThrs)
        for k, k0 in enumerate(k_list):
            for a, a0 in enumerate(a_list):
                E = [self.evalImgs[k0 * A0 * I0 + a0 * I0 + i] for i in i_list]
                E = [e for e in E if e is not None]
                if len(E) == 0:
                    continue
                gt_ig = np.concatenate([e["gtIgnore"] for e in E])
                npig = np.count_nonzero(gt_ig == 0)
                for m, max_det in enumerate(m_list):
                    dt_scores = np.concatenate([e["dtScores"][0:max_det] for e in E])
                    inds = np.argsort(-dt_scores, kind="mergesort")
                    sorted_scores = dt_scores[inds]
                    dtm = np.concatenate([e["dtMatches"][:, 0:max_det] for e in E], axis=1)[:, inds]
                    dt_ig = np.concatenate([e["dtIgnore"][:, 0:max_det] for e in E], axis=1)[:, inds]
                    if npig == 0:
                        continue
                    keep = np.logical_not(dt_ig)
                    tp_sum = np.cumsum(np.logical_and(dtm, keep), axis=1).astype(dtype=float)
                    fp_sum = np.cumsum(np.logical_and(np.logical_not(dtm), keep), axis=1).astype(dtype=float)
                    nd = tp_sum.shape[1]
                    if nd == 0:
                        recall[:, k, a, m] = 0
                        precision[:, :, k, a, m] = 0
                        scores[:, :, k, a, m] = 0
                        continue
                    rc = tp_sum / npig
                    pr = tp_sum / (fp_sum + tp_sum + np.spacing(1))
                    recall[:, k, a, m] = rc[:, -1]
                    pr = np.maximum.accumulate(pr[:, ::-1], axis=1)[:, ::-1]          # the precision envelope
                    for t in range(T):
                        pos = np.searchsorted(rc[t], rec_thrs, side="left")
                        ok = pos < nd
                        pc = np.minimum(pos, nd - 1)
                        precision[t, :, k, a, m] = np.where(ok, pr[t, pc], 0.0)
                        scores[t, :, k, a, m] = np.where(ok, sorted_scores[pc], 0.0)
        self.eval = {"params": p, "counts": [T, R, K, A, M], "date": datetime.datetime.now().strftime("%Y-%m-%d %H:%M:%S"),
                     "precision": precision, "recall": recall, "scores": scores}

    # ------------------------------------------------------------------ cocoeval.py:422-470
    def summarize(self) -> None:
        if not self.eval:
            raise Exception("Please run accumulate() first")
        p = self.params

        def one(ap=1, iouThr=None, areaRng="all", maxDets=100):
            fmt = " {:<18} {} @[ IoU={:<9} | area={:>6s} | maxDets={:>3d} ] = {:0.3f}"
            title, typ = ("Average Precision", "(AP)") if ap == 1 else ("Average Recall", "(AR)")
            iou = "{:0.2f}:{:0.2f}".format(p.iouThrs[0], p.iouThrs[-1]) if iouThr is None else "{:0.2f}".format(iouThr)
            aind = [i for i, l in enumerate(p.areaRngLbl) if l == areaRng]
            mind = [i for i, m in enumerate(p.maxDets) if m == maxDets]
            s = self.eval["precision"] if ap == 1 else self.eval["recall"]
            if iouThr is not None:
                s = s[np.where(iouThr == p.iouThrs)[0]]
            s = s[:, :, :, aind, mind] if ap == 1 else s[:, :, aind, mind]
            mean_s = -1 if len(s[s > -1]) == 0 else np.mean(s[s > -1])
            print(fmt.format(title, typ, iou, areaRng, maxDets, mean_s))
            return mean_s

        md = p.maxDets
        self.stats = np.array([one(1), one(1, iouThr=.5, maxDets=md[2]), one(1, iouThr=.75, maxDets=md[2]),
                               one(1, areaRng="small", maxDets=md[2]), one(1, areaRng="medium", maxDets=md[2]),
                               one(1, areaRng="large", maxDets=md[2]), one(0, maxDets=md[0]), one(0, maxDets=md[1]),
                               one(0, maxDets=md[2]), one(0, areaRng="small", maxDets=md[2]),
                               one(0, areaRng="medium", maxDets=md[2]), one(0, areaRng="large", maxDets=md[2])])

    def __str__(self):
        self.summarize()
        return ""
