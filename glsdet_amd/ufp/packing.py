"""Unified foreground packing (host side, as in the reference: it is a sequential algorithm over at
most a few hundred boxes): enlarge the coarse boxes, merge overlapping ones into foreground
regions, choose a magnification (4 / 2 / 1) per region from the mean object area, and pack the
magnified regions into a near-square canvas with the PH strip-packing heuristic (no rotation,
guillotine cuts) inside a bisection on the strip width.

Follows ufp/UFPMP-Det-Tools/ufp/unified_foreground_packing.py:6-197 (scale_boxes :6-32,
ForegroundRegionGeneration :56-90, Packing :126-177) and ufp/UFPMP-Det-Tools/ufp/spp.py:77-168
(phsppog, recursive_packing).  Arithmetic is float64 like the reference's numpy code, visiting
orders and tie rules are the reference's, so the chip list is identical (tests/test_ufp.py pins it
against the reference's own functions)."""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np

Chip = List[float]      # [src_x, src_y, w, h, canvas_x, canvas_y, magnification]


def enlarge_boxes(boxes: np.ndarray, scale: float, image_wh: Sequence[float]) -> np.ndarray:
    """Each xyxy box grown about its centre by `scale`, clipped to [0, w-1] x [0, h-1]."""
    assert boxes.shape[1] == 4
    half_w = (boxes[:, 2] - boxes[:, 0]) * 0.5 * scale
    half_h = (boxes[:, 3] - boxes[:, 1]) * 0.5 * scale
    cx = (boxes[:, 2] + boxes[:, 0]) * 0.5
    cy = (boxes[:, 3] + boxes[:, 1]) * 0.5
    w, h = image_wh
    out = np.zeros_like(boxes)
    out[:, 0] = np.clip(cx - half_w, 0, w - 1)
    out[:, 2] = np.clip(cx + half_w, 0, w - 1)
    out[:, 1] = np.clip(cy - half_h, 0, h - 1)
    out[:, 3] = np.clip(cy + half_h, 0, h - 1)
    return out


def _magnification(mean_area: float) -> int:
    return 4 if mean_area < 32 * 32 else (2 if mean_area < 96 * 96 else 1)


def foreground_regions(boxes: np.ndarray, grown: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """Greedy single-pass merge: region i swallows every later-or-earlier still-alive region j whose
    union bounding box is smaller than the two areas together; its object-area sum and count follow.
    -> (regions [m,4], magnification [m])."""
    n = boxes.shape[0]
    area_sum = (boxes[:, 2] - boxes[:, 0] + 1) * (boxes[:, 3] - boxes[:, 1] + 1)      # '+1' pixel areas
    count = np.ones(n, dtype=np.int64)
    alive = [True] * n
    for i in range(n):
        if not alive[i]:
            continue
        region = grown[i]
        for j in range(n):
            if j == i or not alive[j]:
                continue
            other = grown[j]
            ux0, uy0 = min(region[0], other[0]), min(region[1], other[1])
            ux1, uy1 = max(region[2], other[2]), max(region[3], other[3])
            both = (region[2] - region[0]) * (region[3] - region[1]) + (other[2] - other[0]) * (other[3] - other[1])
            if (ux1 - ux0) * (uy1 - uy0) < both:
                region = [ux0, uy0, ux1, uy1]
                alive[j] = False
                area_sum[i] += area_sum[j]
                count[i] += count[j]
        grown[i] = region
    mean_area = area_sum / count
    mag = np.array([_magnification(a) for a in mean_area], dtype=np.int64)
    keep = np.array(alive, dtype=bool)
    return grown[keep], mag[keep]


# ---------------------------------------------------------------------------------- strip packing
class _Strip:
    """PH heuristic, 'OG' variant (oriented, guillotine): rectangles sorted by decreasing height;
    the tallest opens a new shelf across the strip, the free area to its right is filled
    recursively by the first rectangle of the best fitting class (exact fit, exact width, exact
    height, strictly smaller)."""

    def __init__(self, width: float, sizes: Sequence[Sequence[float]]):
        self.width = width
        self.sizes = [list(s) for s in sizes]
        self.order = sorted(range(len(sizes)), key=lambda i: -self.sizes[i][1])      # stable, as sorted()
        self.place: List = [None] * len(sizes)
        self.height = 0.0

    def run(self):
        top = 0
        while self.order:
            i = self.order.pop(0)
            w, h = self.sizes[i]
            self.place[i] = (0, top, w, h)
            self._fill(w, top, self.width - w, h)
            top = top + h
        self.height = top
        return top, self.place

    def _fill(self, x, y, w, h):
        rank, best = 6, None
        for i in self.order:
            rw, rh = self.sizes[i]
            if rank > 1 and rw == w and rh == h:
                rank, best = 1, i
                break
            elif rank > 2 and rw == w and rh < h:
                rank, best = 2, i
            elif rank > 3 and rw < w and rh == h:
                rank, best = 3, i
            elif rank > 4 and rw < w and rh < h:
                rank, best = 4, i
            elif rank > 5:
                rank, best = 5, i
        if rank >= 5:
            return
        rw, rh = self.sizes[best]
        self.place[best] = (x, y, rw, rh)
        self.order.remove(best)
        if rank == 2:
            self._fill(x, y + rh, w, h - rh)
        elif rank == 3:
            self._fill(x + rw, y, w - rw, h)
        elif rank == 4:
            smallest = min([min(self.sizes[i][0], self.sizes[i][1]) for i in self.order], default=float("inf"))
            if w - rw < smallest:
                self._fill(x, y + rh, w, h - rh)
            elif h - rh < smallest:
                self._fill(x + rw, y, w - rw, h)
            elif rw < smallest:
                self._fill(x + rw, y, w - rw, rh)
                self._fill(x, y + rh, w, h - rh)
            else:
                self._fill(x, y + rh, rw, h - rh)
                self._fill(x + rw, y, w - rw, h)


def pack_regions(regions: np.ndarray, mag: Sequence[int]) -> Tuple[List[Chip], float, float]:
    """Bisection on the strip width in [300, 2666] for the narrowest strip whose packed height does
    not exceed its width; the layout of the LAST probe is used (as the reference does)."""
    sizes = [[(r[2] - r[0]) * m, (r[3] - r[1]) * m] for r, m in zip(regions, mag)]
    lo, hi = 300, 2666
    placed = []
    while lo <= hi:
        mid = (lo + hi) / 2
        height, placed = _Strip(mid, sizes).run()
        if height > mid:
            lo = mid + 1
        else:
            hi = mid - 1
    pending = [True] * len(sizes)
    chips: List[Chip] = []
    canvas_w = canvas_h = 0
    for (x, y, w, h) in placed:
        canvas_w, canvas_h = max(canvas_w, x + w), max(canvas_h, y + h)
        for i, r in enumerate(regions):               # a placed rectangle claims EVERY pending region of its size
            if not pending[i]:
                continue
            rw, rh = r[2] - r[0], r[3] - r[1]
            if rw * mag[i] == w and rh * mag[i] == h:
                pending[i] = False
                chips.append([r[0], r[1], rw, rh, x, y, mag[i]])
    return chips, canvas_w, canvas_h


def unified_foreground_packing(boxes: np.ndarray, scale: float, input_shape: Sequence[float]):
    """boxes: float ndarray [n,4] xyxy (the coarse detections), input_shape = (image width, height).
    -> (chips, canvas_width, canvas_height); chip = [src_x, src_y, w, h, canvas_x, canvas_y, magnification]."""
    grown = enlarge_boxes(boxes, scale, input_shape)
    regions, mag = foreground_regions(boxes, grown)
    return pack_regions(regions, mag)
