"""Device side of the UFPMP-Det second stage and the two-stage driver (ufp/ufpmp_det_eval.py:253-300):

    coarse detector -> unified foreground packing (host, packing.py) -> mosaic of magnified crops
    (glsdet_ufp_mosaic) -> mmdet test pipeline (glsdet_resize_normalize_pad) -> fine detector ->
    back-mapping + per-class merge NMS (glsdet_ufp_backmap_merge).

Only the chip list (a few hundred floats) and the final detections cross the PCIe bus; the source
image, the mosaic and both network inputs stay in HBM."""
from __future__ import annotations

import ctypes as C
import math
from typing import List, Sequence, Tuple

import numpy as np
import torch

from .. import _lib
from .packing import unified_foreground_packing

MEAN_RGB = (123.675, 116.28, 103.53)            # img_norm_cfg of configs/UFPMP-Det/*.py
STD_RGB = (58.395, 57.12, 57.375)


def rescale_size(old_wh: Tuple[int, int], scale: Tuple[int, int]) -> Tuple[int, int]:
    """mmcv.rescale_size for a (long edge, short edge) tuple: keep the aspect ratio."""
    w, h = old_wh
    f = min(max(scale) / max(h, w), min(scale) / min(h, w))
    return int(w * float(f) + 0.5), int(h * float(f) + 0.5)


class UfpSecondStage:
    def __init__(self, device: str = "cuda:0", img_scale: Tuple[int, int] = (1333, 800), size_divisor: int = 32,
                 mean_rgb: Sequence[float] = MEAN_RGB, std_rgb: Sequence[float] = STD_RGB):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.GlsdetLibraryError("glsdet_amd needs an MI355X visible to PyTorch-ROCm (no CPU fallback)")
        self.device = torch.device(device)
        self.img_scale, self.div = tuple(img_scale), size_divisor
        self._mean = (C.c_double * 3)(*mean_rgb)
        self._std = (C.c_double * 3)(*std_rgb)

    def _stream(self):
        return torch.cuda.current_stream().cuda_stream

    # ---- display_merge_result, ufpmp_det_eval.py:182-193
    def mosaic(self, img_bgr: torch.Tensor, chips: Sequence[Sequence[float]], width: float, height: float) -> torch.Tensor:
        """img_bgr: uint8 [H,W,3] on the device -> fp32 canvas [ceil(height), ceil(width), 3]."""
        assert img_bgr.dtype == torch.uint8 and img_bgr.dim() == 3 and img_bgr.shape[2] == 3 and img_bgr.is_contiguous()
        ch, cw = max(1, math.ceil(height)), max(1, math.ceil(width))
        canvas = torch.empty(ch, cw, 3, dtype=torch.float32, device=self.device)
        n = len(chips)
        # the reference floors every chip field with math.floor on Python floats (ufpmp_det_eval.py:283): floor in
        # float64 on the host, so a value just below an integer cannot round up on its way to float32
        cdev = torch.tensor(np.floor(np.asarray(chips, np.float64)).astype(np.float32).reshape(-1, 7), device=self.device) if n else None
        _lib.check(self.lib.glsdet_ufp_mosaic(img_bgr.data_ptr(), img_bgr.shape[0], img_bgr.shape[1],
                                              cdev.data_ptr() if n else None, n, canvas.data_ptr(), ch, cw, self._stream()),
                   "ufp_mosaic")
        torch.cuda.current_stream().synchronize()
        return canvas

    # ---- Resize(keep_ratio) -> Normalize(to_rgb) -> Pad -> ImageToTensor
    def pipeline_input(self, img_bgr: torch.Tensor):
        """HWC BGR image on the device -> (fp32 [1,3,ph,pw], meta dict as mmdet's img_metas).
        uint8 (a decoded frame, the first stage): cv2's uint8 fixed-point resize, rounded to uint8 before
        Normalize; float32 (the mosaic, a float array in the reference): float bilinear."""
        assert img_bgr.dtype in (torch.float32, torch.uint8) and img_bgr.is_contiguous() and img_bgr.dim() == 3
        h, w = int(img_bgr.shape[0]), int(img_bgr.shape[1])
        nw, nh = rescale_size((w, h), self.img_scale)
        ph, pw = int(math.ceil(nh / self.div)) * self.div, int(math.ceil(nw / self.div)) * self.div
        out = torch.empty(1, 3, ph, pw, dtype=torch.float32, device=self.device)
        fn = self.lib.glsdet_resize_normalize_pad if img_bgr.dtype == torch.float32 else self.lib.glsdet_resize_normalize_pad_u8
        _lib.check(fn(img_bgr.data_ptr(), h, w, nh, nw, out.data_ptr(), ph, pw, self._mean, self._std, self._stream()),
                   "resize_normalize_pad")
        torch.cuda.current_stream().synchronize()
        sf = np.array([nw / w, nh / h, nw / w, nh / h], dtype=np.float32)
        return out, dict(img_shape=(nh, nw, 3), pad_shape=(ph, pw, 3), ori_shape=(h, w, 3), scale_factor=sf, flip=False)

    # ---- back-mapping + merge NMS, ufpmp_det_eval.py:282-300
    def merge(self, dets: torch.Tensor, count: torch.Tensor, chips: Sequence[Sequence[float]], num_classes: int,
              iof_thr: float = 0.9, nms_thr: float = 0.6, max_cand: int = 4096) -> List[np.ndarray]:
        """dets [max_det,7] fp32 / count int32 on the device (the fine detector's buffers for ONE image)
        -> per class ndarray (k,5) x1,y1,x2,y2,score in source-image coordinates, NMS order."""
        max_det = int(dets.shape[0])
        n = len(chips)
        # the reference floors every chip field with math.floor on Python floats (ufpmp_det_eval.py:283): floor in
        # float64 on the host, so a value just below an integer cannot round up on its way to float32
        cdev = torch.tensor(np.floor(np.asarray(chips, np.float64)).astype(np.float32).reshape(-1, 7), device=self.device) if n else None
        ws = torch.zeros(int(self.lib.glsdet_ufp_merge_workspace_bytes(max_cand)) + 256, dtype=torch.uint8, device=self.device)
        off = (-ws.data_ptr()) % 256
        out = torch.zeros(max_cand, 7, dtype=torch.float32, device=self.device)
        cnt = torch.zeros(2, dtype=torch.int32, device=self.device)
        status = torch.zeros(1, dtype=torch.int32, device=self.device)
        _lib.check(self.lib.glsdet_ufp_backmap_merge(dets.data_ptr(), count.data_ptr(), max_det, cdev.data_ptr() if n else None, n,
                                                     iof_thr, nms_thr, max_cand, max_cand, out.data_ptr(), cnt.data_ptr(),
                                                     status.data_ptr(), ws.data_ptr() + off, ws.numel() - off, self._stream()),
                   "ufp_backmap_merge")
        k = int(cnt[0].item())
        if int(status.item()) & 1:
            raise RuntimeError("more than max_cand=%d (chip, detection) matches" % max_cand)
        rows = out[:k].cpu().numpy()
        labels = rows[:, 6].astype(np.int64)
        return [rows[labels == c][:, :5].astype(np.float64) for c in range(num_classes)]


def two_stage_detect(coarse, fine, img_bgr_u8, stage: UfpSecondStage, coarse_cfg: dict, fine_cfg: dict,
                     expand: float = 1.5, use_graph: bool = False):
    """One image through coarse -> UFP -> mosaic -> fine -> merge (ufpmp_det_eval.py:253-300).
    coarse / fine: glsdet_amd.resdet.HipGflDetector; *_cfg: dict(score_thr, iou_thr, nms_pre, max_per_img).
    -> (per class ndarray (k,5) in source-image coordinates, intermediates dict)."""
    img = torch.as_tensor(np.ascontiguousarray(img_bgr_u8)).to(stage.device)
    H, W = int(img.shape[0]), int(img.shape[1])
    x1, m1 = stage.pipeline_input(img)                                # uint8 frame: cv2 uint8 resize
    first = coarse.detect(x1, img_shapes=[m1["img_shape"]], scale_factors=[m1["scale_factor"]], use_graph=use_graph, **coarse_cfg)[0]
    order = np.argsort(first[1], kind="stable")                     # np.concatenate(first_results): class-major
    boxes = first[0][order][:, :4]
    if len(boxes) == 0:
        return [np.zeros((0, 5)) for _ in range(fine.num_classes)], dict(chips=[], first=first)
    chips, cw, ch = unified_foreground_packing(boxes.copy(), expand, [W, H])
    canvas = stage.mosaic(img, chips, cw, ch)
    x2, m2 = stage.pipeline_input(canvas)
    post = dict(score_thr=fine_cfg["score_thr"], iou_thr=fine_cfg["iou_thr"], nms_pre=fine_cfg.get("nms_pre", 1000),
                max_per_img=fine_cfg.get("max_per_img", 500), rescale=True)
    c = fine.compile(1, x2.shape[2], x2.shape[3], post, use_graph=use_graph)
    fine.run(c, x2, torch.tensor([[m2["img_shape"][0], m2["img_shape"][1]]], dtype=torch.float32, device=stage.device),
             torch.tensor(m2["scale_factor"].reshape(1, 4), device=stage.device))
    merged = stage.merge(c.nb["dets"][0], c.nb["count"], chips, fine.num_classes)
    return merged, dict(chips=chips, canvas=canvas, first=first, fine_compiled=c, meta2=m2, canvas_wh=(cw, ch))


class TwoStagePipeline:
    """BASELINE config 5: coarse and fine detector pipelined on separate HIP streams.  `workers` lanes,
    each a pair of host threads with their own streams and their own plan instances (private activation
    buffers): one thread runs preprocessing + coarse detector + packing for its next frame while the other
    runs mosaic + fine detector + merge for its previous one (the host waits of a stage release the GIL,
    the recorded plans are per thread).  Frame i goes to lane i % workers; results come back in order.
    At batch 1 neither detector fills the chip, so several frames in flight is where the throughput is
    (tools/two_stage_bench.py)."""

    def __init__(self, coarse, fine, stage: UfpSecondStage, coarse_cfg: dict, fine_cfg: dict, expand: float = 1.5,
                 depth: int = 2, use_graph: bool = False, workers: int = 1):
        self.coarse, self.fine, self.stage = coarse, fine, stage
        self.coarse_cfg, self.fine_cfg, self.expand, self.depth = coarse_cfg, fine_cfg, expand, depth
        # use_graph: one captured plan per input shape.  Measured (tools/two_stage_bench.py, 540x1024 frames): at batch 1
        # the stages are GPU bound, graph replay gains nothing sequentially (180 frames/s either way) and LOSES the
        # overlap of the two threads (185 vs 280 frames/s with eager plans), so it is off by default.
        self.use_graph = use_graph
        self.workers = max(1, int(workers))

    def _first(self, img_bgr_u8, lane: int = 0):
        st = self.stage
        img = torch.as_tensor(np.ascontiguousarray(img_bgr_u8)).to(st.device)
        H, W = int(img.shape[0]), int(img.shape[1])
        x1, m1 = st.pipeline_input(img)
        first = self.coarse.detect(x1, img_shapes=[m1["img_shape"]], scale_factors=[m1["scale_factor"]], use_graph=self.use_graph,
                                   instance=lane, **self.coarse_cfg)[0]
        order = np.argsort(first[1], kind="stable")
        boxes = first[0][order][:, :4]
        if len(boxes) == 0:
            return img, None
        return img, unified_foreground_packing(boxes.copy(), self.expand, [W, H])

    def _second(self, img, packed, lane: int = 0):
        st, fine = self.stage, self.fine
        if packed is None:
            return [np.zeros((0, 5)) for _ in range(fine.num_classes)]
        chips, cw, ch = packed
        x2, m2 = st.pipeline_input(st.mosaic(img, chips, cw, ch))
        fc = self.fine_cfg
        post = dict(score_thr=fc["score_thr"], iou_thr=fc["iou_thr"], nms_pre=fc.get("nms_pre", 1000),
                    max_per_img=fc.get("max_per_img", 500), rescale=True)
        c = fine.compile(1, x2.shape[2], x2.shape[3], post, use_graph=self.use_graph, instance=lane)
        fine.run(c, x2, torch.tensor([[m2["img_shape"][0], m2["img_shape"][1]]], dtype=torch.float32, device=st.device),
                 torch.tensor(m2["scale_factor"].reshape(1, 4), device=st.device))
        return st.merge(c.nb["dets"][0], c.nb["count"], chips, fine.num_classes)

    def run(self, images: Sequence) -> List[List[np.ndarray]]:
        import queue
        import threading
        results: List = [None] * len(images)
        errors: List[BaseException] = []
        dev = self.stage.device

        def producer(lane, q):
            try:
                with torch.cuda.stream(torch.cuda.Stream(device=dev)):
                    for i in range(lane, len(images), self.workers):
                        item = self._first(images[i], lane)
                        torch.cuda.current_stream().synchronize()
                        q.put((i,) + item)
            except BaseException as e:          # noqa: BLE001 - re-raised in the caller's thread
                errors.append(e)
            finally:
                q.put(None)

        def consumer(lane, q):
            try:
                with torch.cuda.stream(torch.cuda.Stream(device=dev)):
                    while True:
                        item = q.get()
                        if item is None:
                            return
                        i, img, packed = item
                        results[i] = self._second(img, packed, lane)
            except BaseException as e:          # noqa: BLE001
                errors.append(e)
                while q.get() is not None:      # drain so that the producer can finish
                    pass

        ts = []
        for lane in range(self.workers):
            q: "queue.Queue" = queue.Queue(maxsize=self.depth)
            ts += [threading.Thread(target=producer, args=(lane, q)), threading.Thread(target=consumer, args=(lane, q))]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        if errors:
            raise errors[0]
        return results
