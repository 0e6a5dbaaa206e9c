"""Device-side image preprocessing of the drone flavour:
`resize_image` (PIL BICUBIC, optional letterbox on a gray canvas) + `preprocess_input` + HWC->CHW
(drone/models/core/utils.py:21-34,46-50; drone/yolo.py:125-134), bit-identical to the reference's
CPU result.  Host side here = the coefficient tables Pillow's resampler would compute; the two
resampling passes, the normalisation and the layout change run in libglsdet_hip
(glsdet_pil_resize_normalize)."""
from __future__ import annotations

import ctypes as C
from typing import Dict, Sequence, Tuple

import numpy as np
import torch

from . import _lib

MEAN = (0.485, 0.456, 0.406)          # utils.py:48
STD = (0.229, 0.224, 0.225)           # utils.py:49
GRAY = 128                            # letterbox canvas colour, utils.py:30
_BITS = 22                            # Pillow PRECISION_BITS = 32 - 8 - 2


def _bicubic(x: np.ndarray) -> np.ndarray:
    """Pillow's bicubic_filter (a = -0.5), evaluated in float64 like the C code."""
    a = -0.5
    x = np.abs(x)
    return np.where(x < 1.0, ((a + 2.0) * x - (a + 3.0)) * x * x + 1,
                    np.where(x < 2.0, (((x - 5) * x + 8) * x - 4) * a, 0.0))


def pil_bicubic_tables(in_size: int, out_size: int) -> Tuple[np.ndarray, np.ndarray, int]:
    """Pillow's precompute_coeffs + normalize_coeffs_8bpc for a whole-image box.
    -> (bounds int32 [out,2] = (first index, tap count), kk int32 [out,ksize] = round(w * 2^22), ksize)."""
    scale = float(in_size) / out_size
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)              # C (int) cast: truncation
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = _bicubic((np.arange(xmax) + xmin - center + 0.5) * ss)
        ww = 0.0
        for v in w:                                            # same left-to-right sum as the C loop
            ww += float(v)
        if ww != 0.0:
            w = w / ww
        fixed = w * float(1 << _BITS)
        kk[xx, :xmax] = np.where(fixed < 0, (-0.5 + fixed), (0.5 + fixed)).astype(np.int64).astype(np.int32)
        bounds[xx] = (xmin, xmax)
    return bounds, kk, ksize


class DronePreprocessor:
    """images (uint8 HWC RGB, any sizes) -> float32 [B,3,H,W] on the GPU, as yolo.py:125-134 does on the
    CPU for one image.  Tables are cached per (in, out) size."""

    def __init__(self, device: str = "cuda:0"):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.GlsdetLibraryError("glsdet_amd needs an MI355X visible to PyTorch-ROCm (no CPU fallback)")
        self.device = torch.device(device)
        self._tables: Dict[Tuple[int, int], tuple] = {}
        self._mean = (C.c_double * 3)(*MEAN)
        self._std = (C.c_double * 3)(*STD)
        gray = np.full(3, GRAY, np.float32) / np.float32(255.0)                  # the same mixed arithmetic
        gray = (gray.astype(np.float64) - np.array(MEAN)).astype(np.float32)
        self._gray = torch.from_numpy((gray.astype(np.float64) / np.array(STD)).astype(np.float32))

    def _table(self, n_in: int, n_out: int):
        key = (n_in, n_out)
        if key not in self._tables:
            b, k, ks = pil_bicubic_tables(n_in, n_out)
            self._tables[key] = (torch.from_numpy(b).to(self.device), torch.from_numpy(k).to(self.device), ks)
        return self._tables[key]

    def __call__(self, images: Sequence, input_shape: Sequence[int], letterbox_image: bool = False,
                 out: torch.Tensor = None) -> torch.Tensor:
        H, W = int(input_shape[0]), int(input_shape[1])
        n = len(images)
        if out is None:
            out = torch.empty(n, 3, H, W, dtype=torch.float32, device=self.device)
        assert out.shape == (n, 3, H, W) and out.dtype == torch.float32 and out.is_contiguous()
        if letterbox_image:
            out.copy_(self._gray.to(self.device).view(1, 3, 1, 1).expand(n, 3, H, W))
        keep = []
        for b, img in enumerate(images):
            src = torch.as_tensor(np.ascontiguousarray(img)) if not isinstance(img, torch.Tensor) else img
            if src.dtype != torch.uint8 or src.dim() != 3 or src.shape[2] != 3:
                raise ValueError("images must be uint8 HWC RGB arrays")
            src = src.to(self.device).contiguous()
            ih, iw = int(src.shape[0]), int(src.shape[1])
            if letterbox_image:                                                     # utils.py:24-31
                scale = min(W / iw, H / ih)
                nw, nh = int(iw * scale), int(ih * scale)
                ox, oy = (W - nw) // 2, (H - nh) // 2
            else:
                nw, nh, ox, oy = W, H, 0, 0
            xb, xk, xks = self._table(iw, nw)
            yb, yk, yks = self._table(ih, nh)
            tmp = torch.empty(ih * nw * 3, dtype=torch.uint8, device=self.device)
            keep += [src, tmp]
            _lib.check(self.lib.glsdet_pil_resize_normalize(
                src.data_ptr(), ih, iw, xb.data_ptr(), xk.data_ptr(), xks, nw, yb.data_ptr(), yk.data_ptr(), yks, nh,
                tmp.data_ptr(), out[b].data_ptr(), H, W, oy, ox, self._mean, self._std,
                torch.cuda.current_stream().cuda_stream), "pil_resize_normalize")
        torch.cuda.current_stream().synchronize()       # `keep` may be released after this
        return out
