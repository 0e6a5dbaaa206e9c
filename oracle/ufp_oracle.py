"""CPU oracle of the UFPMP-Det second stage  --  TEST INFRASTRUCTURE ONLY.

Restates, in numpy, the steps of ufp/ufpmp_det_eval.py between the two detector calls and after the
second one:

    display_merge_result   :182-193   mosaic of magnified crops (cv2.resize, INTER_LINEAR, uint8)
    the mmdet test pipeline            Resize(keep_ratio, (1333,800)) -> Normalize(to_rgb) -> Pad(32)
                                       (mmdet/datasets/pipelines/transforms.py:30,671,572 over mmcv)
    compute_iof            :36-50,    back-mapping of the fine detections into the source image
    main                   :282-296
    py_cpu_nms             :149-178   per-class merge NMS ('+1' areas, keep while IoU <= thr)

PARITY UNPINNED for the image steps: cv2 and mmcv are not importable here and the reference holds no
fixture.  `cv2_resize_linear_u8` restates OpenCV's uint8 bilinear resize from its published
implementation (11-bit fixed-point coefficients, the `(b0*(S0>>4))>>16` vertical pass); the float
resize is the plain half-pixel bilinear formula.  The box arithmetic is the reference's formulas in
float64 (the reference mixes numpy float32 scalars and Python ints; which of the two wins depends on
the numpy version it ran under)."""
from __future__ import annotations

import math
from typing import List, Sequence, Tuple

import numpy as np


# --------------------------------------------------------------------------- cv2-style resizing
def _linear_taps(dst: int, src: int):
    """OpenCV's per-axis setup for INTER_LINEAR: source index, fraction (float32), and whether the
    second tap exists."""
    scale = 1.0 / (float(dst) / float(src))
    idx = np.empty(dst, np.int64)
    frac = np.empty(dst, np.float32)
    for d in range(dst):
        f = np.float32((d + 0.5) * scale - 0.5)
        s = int(math.floor(f))
        f = np.float32(f - s)
        if s < 0:
            s, f = 0, np.float32(0)
        if s >= src - 1:
            s, f = src - 1, np.float32(0)
        idx[d], frac[d] = s, f
    return idx, frac


def cv2_resize_linear_u8(src: np.ndarray, dw: int, dh: int) -> np.ndarray:
    """cv2.resize(src, (dw, dh)) for uint8 HWC, INTER_LINEAR (an equal size returns a copy)."""
    sh, sw = src.shape[:2]
    if (sw, sh) == (dw, dh):
        return src.copy()
    xi, xf = _linear_taps(dw, sw)
    yi, yf = _linear_taps(dh, sh)
    rnd = lambda v: np.rint(v.astype(np.float64) * 2048.0).astype(np.int64)      # saturate_cast<short>(coef * 2048)
    a0, a1 = rnd(np.float32(1) - xf), rnd(xf)
    b0, b1 = rnd(np.float32(1) - yf), rnd(yf)
    s = src.astype(np.int64)
    x1 = np.minimum(xi + 1, sw - 1)
    rows = s[:, xi] * a0[None, :, None] + s[:, x1] * a1[None, :, None]             # [sh, dw, c], scale 2048
    y1 = np.minimum(yi + 1, sh - 1)
    out = (((b0[:, None, None] * (rows[yi] >> 4)) >> 16) + ((b1[:, None, None] * (rows[y1] >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def display_merge_result(img_bgr: np.ndarray, chips: Sequence[Sequence[float]], w: float, h: float) -> np.ndarray:
    """ufpmp_det_eval.py:182-193 -> float64 canvas [ceil(h), ceil(w), 3], zeros outside the chips."""
    w, h = math.ceil(w), math.ceil(h)
    canvas = np.zeros((h, w, 3))
    for chip in chips:
        x1, y1, cw, ch, nx, ny, s = [math.floor(v) for v in chip]
        if cw == 0 or ch == 0:
            continue
        crop = img_bgr[y1:y1 + ch, x1:x1 + cw, :]
        canvas[ny:ny + ch * s, nx:nx + cw * s, :] = cv2_resize_linear_u8(crop, cw * s, ch * s)
    return canvas


def rescale_size(old_wh: Tuple[int, int], scale: Tuple[int, int]) -> Tuple[Tuple[int, int], float]:
    """mmcv.rescale_size with a (long, short) tuple scale."""
    w, h = old_wh
    f = min(max(scale) / max(h, w), min(scale) / min(h, w))
    return (int(w * float(f) + 0.5), int(h * float(f) + 0.5)), f


def resize_linear_f(src: np.ndarray, dw: int, dh: int) -> np.ndarray:
    """half-pixel bilinear resize of a float image (cv2.resize INTER_LINEAR semantics: edge clamp)."""
    sh, sw = src.shape[:2]
    xi, xf = _linear_taps(dw, sw)
    yi, yf = _linear_taps(dh, sh)
    x1, y1 = np.minimum(xi + 1, sw - 1), np.minimum(yi + 1, sh - 1)
    xf, yf = xf.astype(np.float64), yf.astype(np.float64)
    rows = src[:, xi] * (1.0 - xf)[None, :, None] + src[:, x1] * xf[None, :, None]
    return rows[yi] * (1.0 - yf)[:, None, None] + rows[y1] * yf[:, None, None]


MEAN = np.array([123.675, 116.28, 103.53])          # configs/UFPMP-Det/*.py img_norm_cfg (RGB order)
STD = np.array([58.395, 57.12, 57.375])


def mmdet_test_pipeline(img_bgr: np.ndarray, img_scale=(1333, 800), size_divisor: int = 32):
    """Resize(keep_ratio) -> Normalize(mean, std, to_rgb=True) -> Pad(size_divisor) -> CHW float32.
    (An exact 2x downscale, which OpenCV silently runs as INTER_AREA, is not special-cased.)
    -> (tensor [1,3,H,W], meta dict(img_shape, pad_shape, scale_factor))."""
    h, w = img_bgr.shape[:2]
    (nw, nh), _ = rescale_size((w, h), img_scale)
    if img_bgr.dtype == np.uint8:        # a decoded frame: cv2's uint8 resize, rounded to uint8 before Normalize
        img = cv2_resize_linear_u8(img_bgr, nw, nh).astype(np.float64)
    else:                                # the mosaic is a float array in the reference: float bilinear
        img = resize_linear_f(img_bgr.astype(np.float64), nw, nh)
    scale_factor = np.array([nw / w, nh / h, nw / w, nh / h], dtype=np.float32)
    rgb = img[:, :, ::-1].astype(np.float32)
    rgb = (rgb.astype(np.float64) - MEAN).astype(np.float32)
    rgb = (rgb.astype(np.float64) * (1.0 / STD)).astype(np.float32)
    ph, pw = int(np.ceil(nh / size_divisor)) * size_divisor, int(np.ceil(nw / size_divisor)) * size_divisor
    out = np.zeros((1, 3, ph, pw), np.float32)
    out[0, :, :nh, :nw] = rgb.transpose(2, 0, 1)
    return out, dict(img_shape=(nh, nw, 3), pad_shape=(ph, pw, 3), scale_factor=scale_factor, ori_shape=(h, w, 3))


# --------------------------------------------------------------------------- boxes
def compute_iof(a, b) -> float:
    """ufpmp_det_eval.py:36-50: intersection over the SMALLER of the two areas."""
    l, t, r, d = max(a[0], b[0]), max(a[1], b[1]), min(a[2], b[2]), min(a[3], b[3])
    if l >= r or t >= d:
        return 0.0
    return (r - l) * (d - t) / min((a[2] - a[0]) * (a[3] - a[1]), (b[2] - b[0]) * (b[3] - b[1]))


def py_cpu_nms(dets: np.ndarray, thresh: float) -> List[int]:
    """ufpmp_det_eval.py:149-178: greedy, '+1' pixel areas, a box survives while IoU <= thresh.
    Ties in score: the reference's `argsort()[::-1]` order (for equal scores, higher index first)."""
    x1, y1, x2, y2, sc = dets[:, 0], dets[:, 1], dets[:, 2], dets[:, 3], dets[:, 4]
    areas = (x2 - x1 + 1) * (y2 - y1 + 1)
    order = np.argsort(sc, kind="stable")[::-1]
    keep = []
    while order.size > 0:
        i = order[0]
        keep.append(int(i))
        xx1, yy1 = np.maximum(x1[i], x1[order[1:]]), np.maximum(y1[i], y1[order[1:]])
        xx2, yy2 = np.minimum(x2[i], x2[order[1:]]), np.minimum(y2[i], y2[order[1:]])
        inter = np.maximum(0.0, xx2 - xx1 + 1) * np.maximum(0.0, yy2 - yy1 + 1)
        ovr = inter / (areas[i] + areas[order[1:]] - inter)
        order = order[np.where(ovr <= thresh)[0] + 1]
    return keep


def map_back_and_merge(second_results: Sequence[np.ndarray], chips: Sequence[Sequence[float]], num_classes: int = 10,
                       iof_thr: float = 0.9, nms_thr: float = 0.6) -> List[np.ndarray]:
    """ufpmp_det_eval.py:278-300: a fine detection that lies (IoF > 0.9) in a chip's canvas rectangle
    is mapped back through that chip's magnification and offset; per class py_cpu_nms(0.6).
    second_results: per class ndarray (n,5) in mosaic coordinates -> per class ndarray (k,5) kept, in
    NMS order, source-image coordinates (float64)."""
    mapped: List[list] = [[] for _ in range(num_classes)]
    for chip in chips:
        ox, oy, w, h, nx, ny, s = [math.floor(v) for v in chip]
        rect = [nx, ny, nx + w * s, ny + h * s]
        for c, dets in enumerate(second_results):
            for d in np.asarray(dets, np.float64).reshape(-1, 5):
                if compute_iof(d[:4], rect) > iof_thr:
                    bw, bh = (d[2] - d[0]) / s, (d[3] - d[1]) / s
                    bx, by = (d[0] - nx) / s + ox, (d[1] - ny) / s + oy
                    mapped[c].append([bx, by, bx + bw, by + bh, d[4]])
    out = []
    for c in range(num_classes):
        r = np.asarray(mapped[c], np.float64).reshape(-1, 5)
        out.append(r[py_cpu_nms(r, nms_thr)] if len(r) else r)
    return out
