"""Import-name shim (see ../README.md): the two cv2 calls of display_merge_result (ufpmp_det_eval.py:182-193).

`imread` decodes with Pillow.  `resize` serves what the script asks for -- a uint8 crop magnified by an integer factor
(1, 2 or 4: UFP's zoom levels) with cv2's default INTER_LINEAR -- on the device, through the same kernel that
composes whole mosaics (glsdet_ufp_mosaic: cv2's 11-bit fixed-point arithmetic)."""
import numpy as np

INTER_LINEAR = 1


def imread(path, flags=1):
    from PIL import Image
    try:
        with Image.open(str(path)) as im:
            return np.ascontiguousarray(np.asarray(im.convert("RGB"))[:, :, ::-1])
    except (FileNotFoundError, OSError):
        return None                       # cv2.imread's convention


def resize(src, dsize, dst=None, fx=0, fy=0, interpolation=INTER_LINEAR):
    import torch
    from glsdet_amd.ufp import UfpSecondStage
    dw, dh = int(dsize[0]), int(dsize[1])
    h, w = src.shape[:2]
    if interpolation != INTER_LINEAR or src.dtype != np.uint8 or src.ndim != 3 or src.shape[2] != 3 or w == 0 or h == 0 or \
            dw % w or dh % h or dw // w != dh // h:
        raise NotImplementedError("cv2.resize shim: uint8 BGR crops magnified by one integer factor, INTER_LINEAR")
    stage = resize._stage = getattr(resize, "_stage", None) or UfpSecondStage()
    img = torch.from_numpy(np.ascontiguousarray(src)).to(stage.device)
    canvas = stage.mosaic(img, [[0, 0, w, h, 0, 0, dw // w]], dw, dh)
    return canvas.cpu().numpy().astype(np.uint8)
