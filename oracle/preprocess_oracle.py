"""CPU oracle of the drone flavour's image preprocessing  --  TEST INFRASTRUCTURE ONLY.

The reference's algorithm here IS "call Pillow, then three numpy in-place ops"
(drone/models/core/utils.py:21-34,46-50; drone/yolo.py:125-134), so the restatement calls the
same Pillow entry point.  Pinned by tests/golden/preprocess_golden.npz, produced by running the
reference's own `resize_image` / `preprocess_input` in the build container (Pillow 12.2.0)."""
import numpy as np
from PIL import Image


def resize_image(image: Image.Image, size, letterbox_image: bool) -> Image.Image:
    """utils.py:21-34: size = (w, h); letterbox keeps the aspect ratio on a (128,128,128) canvas."""
    iw, ih = image.size
    w, h = size
    if not letterbox_image:
        return image.resize((w, h), Image.BICUBIC)
    scale = min(w / iw, h / ih)
    nw, nh = int(iw * scale), int(ih * scale)
    canvas = Image.new("RGB", size, (128, 128, 128))
    canvas.paste(image.resize((nw, nh), Image.BICUBIC), ((w - nw) // 2, (h - nh) // 2))
    return canvas


def preprocess_input(image: np.ndarray) -> np.ndarray:
    """utils.py:46-50, in place on a float32 array (float64 constants: numpy computes each step in
    double and rounds back to float32)."""
    image /= 255.0
    image -= np.array([0.485, 0.456, 0.406])
    image /= np.array([0.229, 0.224, 0.225])
    return image


def drone_preprocess(img_u8: np.ndarray, input_shape, letterbox_image: bool) -> np.ndarray:
    """yolo.py:125-134 for one uint8 HWC RGB array -> float32 [1,3,H,W]."""
    pil = resize_image(Image.fromarray(img_u8, "RGB"), (input_shape[1], input_shape[0]), letterbox_image)
    return np.expand_dims(np.transpose(preprocess_input(np.array(pil, dtype="float32")), (2, 0, 1)), 0)
