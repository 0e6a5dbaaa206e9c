"""Import-name shim (see ../README.md): `mmcv` as far as ufpmp_det_eval.py uses it (`mmcv.imread`, :93)."""
import numpy as np


def imread(img_or_path, flag="color"):
    """BGR uint8 [H,W,3] like cv2.imread / mmcv.imread (decoded with Pillow; an ndarray passes through)."""
    if isinstance(img_or_path, np.ndarray):
        return img_or_path
    from PIL import Image
    with Image.open(str(img_or_path)) as im:
        return np.ascontiguousarray(np.asarray(im.convert("RGB"))[:, :, ::-1])
