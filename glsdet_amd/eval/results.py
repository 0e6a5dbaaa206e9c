"""Result hand-off formats of the reference and the file-level merge built on them.

    detection-results/<image>.txt     drone/yolo.py:296-303   "<class> <score[:6]> <left> <top> <right> <bottom>"
                                      parsers: ufp/myufp_eval.py:27-76, drone/merge_results.py:20-39
    COCO result records               ufp/ufpmp_det_eval.py:303-326  (int-truncated corners -> xywh)
    merge of two result directories   drone/merge_results.py:132-172 (concatenate, batched_nms 0.65, rewrite)

The merge NMS runs on the GPU through `glsdet_nms` (class-aware greedy NMS, IoU > thr suppresses, areas
without +1 -- torchvision.ops.boxes.batched_nms).  The lossy steps of the reference are kept: the score
STRING cut to 6 characters, coordinates through int()."""
from __future__ import annotations

import os
from typing import Dict, Iterable, List, Sequence, Tuple

import numpy as np
import torch

VISDRONE_CLASSES = ("pedestrian", "people", "bicycle", "car", "van", "truck", "tricycle", "awning-tricycle", "bus", "motor")


# ------------------------------------------------------------------------------------ txt lines
def detection_line(class_name: str, score, left, top, right, bottom) -> str:
    """drone/yolo.py:302-303: the score is str()-ed and cut to 6 characters, the corners go through int()."""
    return "%s %s %s %s %s %s\n" % (class_name, str(score)[:6], str(int(left)), str(int(top)), str(int(right)), str(int(bottom)))


def write_detection_results(path: str, dets: np.ndarray, class_names: Sequence[str]) -> int:
    """dets: the drone NMS output for one image, rows [top, left, bottom, right, obj, cls_conf, cls_id]
    (non_max_suppression after yolo_correct_boxes, utils_bbox.py:31,419), or None for "no detection"
    (the reference then writes an empty file, yolo.py:286-287).  -> number of lines written."""
    n = 0
    with open(path, "w") as f:
        if dets is None:
            return 0
        for row in np.asarray(dets):
            top, left, bottom, right = row[:4]
            score = row[4] * row[5]                        # same dtype arithmetic as the reference (float32 rows)
            f.write(detection_line(class_names[int(np.int32(row[6]))], score, left, top, right, bottom))
            n += 1
    return n


def parse_detection_results(path: str, class_index: Dict[str, int], prob: bool = True) -> List[List[float]]:
    """drone/merge_results.py:20-39: rows [left, top, right, bottom, score, class index] in file order
    (prob=False: ground-truth files without the score column, score 1)."""
    rows = []
    with open(path, "r", encoding="utf-8") as f:
        for line in f:
            info = line[:-1].split(" ") if line.endswith("\n") else line.split(" ")
            if len(info) < (6 if prob else 5):
                continue
            if prob:
                r = [float(info[i]) for i in range(2, 6)] + [float(info[1])]
            else:
                r = [float(info[i]) for i in range(1, 5)] + [1]
            rows.append(r + [class_index[info[0]]])
    return rows


def parse_per_class(path: str, class_index: Dict[str, int], prob: bool = True, min_score: float = None) -> List[List[List[float]]]:
    """ufp/myufp_eval.py:55-76 (`get_annotations`) / :27-53 (`get_annotationsFirst`, min_score=0.2 and
    empty classes dropped by the caller): per class a list of [left, top, right, bottom, score]."""
    out: List[List[List[float]]] = [[] for _ in range(len(class_index))]
    for r in parse_detection_results(path, class_index, prob):
        if min_score is None or r[4] > min_score:
            out[int(r[5])].append(r[:5])
    return out


# ------------------------------------------------------------------------------------ COCO records
def coco_records(image_id, per_class: Sequence[np.ndarray]) -> List[dict]:
    """ufp/ufpmp_det_eval.py:303-324: per class rows (x1, y1, x2, y2, score) -> result dicts with the corners
    cut by int() and the box as [x1, y1, x2-x1, y2-y1]; category_id is the class index."""
    out = []
    for c, rows in enumerate(per_class):
        for x1, y1, x2, y2, score in np.asarray(rows, np.float64).reshape(-1, 5):
            x1, y1, x2, y2 = int(x1), int(y1), int(x2), int(y2)
            out.append({"image_id": image_id, "category_id": c, "score": float(score), "bbox": [x1, y1, x2 - x1, y2 - y1]})
    return out


# ------------------------------------------------------------------------------------ merging result files
class ResultMerger:
    """drone/merge_results.py:132-172 for one image at a time, NMS on the device."""

    def __init__(self, classes: Sequence[str] = VISDRONE_CLASSES, nms_thres: float = 0.65, device: str = "cuda:0",
                 capacity: int = 8192):
        from ..engine import Engine
        self.classes = tuple(classes)
        self.index = {c: i for i, c in enumerate(self.classes)}
        self.thr, self.cap = float(nms_thres), int(capacity)
        self.eng = Engine(device=device, dtype="f32")
        self._nb = self.eng.nms_buffers(1, self.cap, self.cap, self.cap)
        self._pred = torch.zeros(1, self.cap, 5 + len(self.classes), dtype=torch.float32, device=self.eng.device)

    def merge_rows(self, rows: np.ndarray) -> np.ndarray:
        """rows [n,6] left, top, right, bottom, score, class -> the kept rows in descending score order."""
        rows = np.asarray(rows, np.float32).reshape(-1, 6)
        n, nc = len(rows), len(self.classes)
        if n == 0:
            return rows
        if n > self.cap:
            raise RuntimeError("more rows (%d) than the merger's capacity %d" % (n, self.cap))
        # obj = 1 and the class scores are -1 everywhere except the row's own class: the class max is then the row's
        # score at the row's class also for a score of exactly 0 (a '0.0000' cut of the reference's 6-character score
        # string, which merge_results.py keeps), and the all -1 padding rows fall under the threshold of -0.5
        host = np.full((self.cap, 5 + nc), -1.0, np.float32)
        host[:, :4] = 0.0
        host[:, 4] = 1.0
        host[:n, :4] = rows[:, :4]
        host[np.arange(n), 5 + rows[:, 5].astype(np.int64)] = rows[:, 4]
        self._pred[0].copy_(torch.from_numpy(host))
        dets, count, status = self.eng.nms(self._pred, nc, 1, -0.5, self.thr, self._nb)
        torch.cuda.current_stream(self.eng.device).synchronize()
        if int(status.item()) & 1:
            raise RuntimeError("NMS candidate capacity exceeded")
        k = int(count[0].item())
        d = dets[0, :k].cpu().numpy()
        return np.concatenate([d[:, :4], (d[:, 4] * d[:, 5])[:, None], d[:, 6:7]], axis=1)

    def merge_files(self, paths: Iterable[str], out_path: str) -> int:
        rows = [r for p in paths if os.path.exists(p) for r in parse_detection_results(p, self.index)]
        kept = self.merge_rows(np.asarray(rows, np.float32).reshape(-1, 6))
        with open(out_path, "w") as f:
            for r in kept:
                # merge_results.py:166-167: the score is written in full here (float()), corners through int()
                f.write("%s %s %s %s %s %s\n" % (self.classes[int(r[5])], float(r[4]), int(r[0]), int(r[1]), int(r[2]), int(r[3])))
        return len(kept)

    def merge_dirs(self, dirs: Sequence[str], out_dir: str) -> int:
        os.makedirs(out_dir, exist_ok=True)
        total = 0
        for name in sorted(os.listdir(dirs[0])):
            total += self.merge_files([os.path.join(d, name) for d in dirs], os.path.join(out_dir, name))
        return total
