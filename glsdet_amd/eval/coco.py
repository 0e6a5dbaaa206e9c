"""Minimal COCO annotation container: the part of `pycocotools.coco.COCO` the evaluation path of the
reference touches (ufp/ufpmp_det_eval.py:226-236, 328-331: `COCO(ann_file)`, `getImgIds`, `loadImgs`,
`loadRes(result_json)`) plus the accessors COCOeval calls (drone/models/core/cocoeval.py:95-101).
bbox only: no masks, no keypoints, no captions."""
from __future__ import annotations

import copy
import json
from collections import defaultdict
from typing import Iterable, List, Sequence, Union

import numpy as np


def _as_list(v) -> list:
    return list(v) if isinstance(v, (list, tuple, set, np.ndarray)) else [v]


class COCO:
    def __init__(self, annotation_file: Union[str, dict, None] = None):
        self.dataset, self.anns, self.cats, self.imgs = dict(), dict(), dict(), dict()
        self.imgToAnns, self.catToImgs = defaultdict(list), defaultdict(list)
        if annotation_file is not None:
            if isinstance(annotation_file, dict):
                self.dataset = annotation_file
            else:
                with open(annotation_file, "r") as f:
                    self.dataset = json.load(f)
            if not isinstance(self.dataset, dict):
                raise TypeError("annotation file format %s not supported" % type(self.dataset))
            self.createIndex()

    def createIndex(self) -> None:
        anns, cats, imgs = {}, {}, {}
        img_to_anns, cat_to_imgs = defaultdict(list), defaultdict(list)
        for ann in self.dataset.get("annotations", []):
            img_to_anns[ann["image_id"]].append(ann)
            anns[ann["id"]] = ann
        for img in self.dataset.get("images", []):
            imgs[img["id"]] = img
        for cat in self.dataset.get("categories", []):
            cats[cat["id"]] = cat
        if "categories" in self.dataset:
            for ann in self.dataset.get("annotations", []):
                cat_to_imgs[ann["category_id"]].append(ann["image_id"])
        self.anns, self.cats, self.imgs, self.imgToAnns, self.catToImgs = anns, cats, imgs, img_to_anns, cat_to_imgs

    # ---- accessors
    def getAnnIds(self, imgIds=[], catIds=[], areaRng=[], iscrowd=None) -> List[int]:
        imgIds, catIds = _as_list(imgIds), _as_list(catIds)
        if len(imgIds) == len(catIds) == len(areaRng) == 0:
            anns = self.dataset["annotations"]
        else:
            if len(imgIds) > 0:
                anns = [a for i in imgIds if i in self.imgToAnns for a in self.imgToAnns[i]]
            else:
                anns = self.dataset["annotations"]
            if len(catIds) > 0:
                cs = set(catIds)
                anns = [a for a in anns if a["category_id"] in cs]
            if len(areaRng) > 0:
                anns = [a for a in anns if areaRng[0] < a["area"] < areaRng[1]]
        if iscrowd is not None:
            return [a["id"] for a in anns if a["iscrowd"] == iscrowd]
        return [a["id"] for a in anns]

    def getCatIds(self) -> List[int]:
        return [c["id"] for c in self.dataset["categories"]]

    def getImgIds(self) -> List[int]:
        return list(self.imgs.keys())

    def loadAnns(self, ids=[]) -> List[dict]:
        return [self.anns[i] for i in ids] if isinstance(ids, (list, tuple, np.ndarray)) else [self.anns[ids]]

    def loadCats(self, ids=[]) -> List[dict]:
        return [self.cats[i] for i in ids] if isinstance(ids, (list, tuple, np.ndarray)) else [self.cats[ids]]

    def loadImgs(self, ids=[]) -> List[dict]:
        return [self.imgs[i] for i in ids] if isinstance(ids, (list, tuple, np.ndarray)) else [self.imgs[ids]]

    # ---- results
    def loadRes(self, resFile: Union[str, Sequence[dict], np.ndarray]) -> "COCO":
        """bbox results -> COCO object: a json path, a list of records (image_id, category_id, bbox xywh,
        score) or an ndarray [N,7] (image_id, x, y, w, h, score, category_id).  Ids are assigned from 1,
        area = w*h, iscrowd = 0, like pycocotools."""
        res = COCO()
        res.dataset["images"] = [img for img in self.dataset["images"]]
        if isinstance(resFile, str):
            with open(resFile) as f:
                anns = json.load(f)
        elif isinstance(resFile, np.ndarray):
            assert resFile.ndim == 2 and resFile.shape[1] == 7
            anns = [dict(image_id=int(r[0]), bbox=[r[1], r[2], r[3], r[4]], score=r[5], category_id=int(r[6])) for r in resFile.tolist()]
        else:
            anns = copy.deepcopy(list(resFile))
        assert isinstance(anns, list), "results in not an array of objects"
        assert set(a["image_id"] for a in anns) == (set(a["image_id"] for a in anns) & set(self.getImgIds())), \
            "Results do not correspond to current coco set"
        if len(anns) and "bbox" not in anns[0]:
            raise ValueError("only bbox results are supported")
        res.dataset["categories"] = copy.deepcopy(self.dataset["categories"])
        for i, ann in enumerate(anns):
            bb = ann["bbox"]
            ann["area"] = bb[2] * bb[3]
            ann["id"] = i + 1
            ann["iscrowd"] = 0
        res.dataset["annotations"] = anns
        res.createIndex()
        return res
