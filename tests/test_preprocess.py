"""image preprocessing of the drone flavour (PIL BICUBIC resize, optional
letterbox, preprocess_input, HWC->CHW).  Goldens come from the reference's own `resize_image` /
`preprocess_input` (tests/golden/make_golden.py::preprocess_cases).  Everything is BIT-EXACT:
uint8 resampling with Pillow's fixed-point arithmetic, then numpy's mixed f32/f64 normalisation."""
import os

import numpy as np
import pytest
import torch

from oracle import preprocess_oracle as P

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = [((77, 123), (64, 96), False), ((77, 123), (64, 96), True), ((150, 90), (64, 96), True),
         ((64, 96), (64, 96), False), ((300, 500), (64, 96), False), ((31, 45), (64, 96), True),
         ((97, 64), (96, 64), True)]


def synth_image(shape, seed):
    h, w = shape
    rng = np.random.default_rng([seed, 0x1A6E])
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    img = np.stack([127 + 120 * np.sin(xx / (5 + 3 * c) + seed) * np.cos(yy / (7 + 2 * c)) for c in range(3)], -1)
    img += rng.normal(0, 25, img.shape)
    img[: h // 5, : w // 4] = 255
    img[-(h // 6):, -(w // 5):] = 0
    return np.clip(img, 0, 255).astype(np.uint8)


@pytest.fixture(scope="module")
def pre_golden():
    return np.load(os.path.join(ROOT, "tests", "golden", "preprocess_golden.npz"))


@pytest.mark.parametrize("i", range(len(CASES)))
def test_oracle_equals_the_reference_bit_for_bit(pre_golden, i):
    ishape, shape, lb = CASES[i]
    got = P.drone_preprocess(synth_image(ishape, i), shape, lb)
    want = pre_golden["pre/%d" % i]
    assert got.dtype == np.float32 and got.shape == want.shape
    assert np.array_equal(got, want)


def _two_pass_numpy(img, out_hw):
    """The device kernels' arithmetic in numpy int64, driven by the product's host tables."""
    from glsdet_amd.preprocess import pil_bicubic_tables
    h, w = img.shape[:2]
    oh, ow = out_hw

    def one_pass(a, n_out):          # resample axis 1 of [rows, n_in, 3]
        b, k, _ = pil_bicubic_tables(a.shape[1], n_out)
        out = np.empty((a.shape[0], n_out, 3), np.uint8)
        for xx in range(n_out):
            x0, cnt = b[xx]
            s = (1 << 21) + (a[:, x0:x0 + cnt].astype(np.int64) * k[xx, :cnt].astype(np.int64)[None, :, None]).sum(1)
            out[:, xx] = np.clip(s >> 22, 0, 255)
        return out
    t = one_pass(img, ow)
    return one_pass(t.transpose(1, 0, 2), oh).transpose(1, 0, 2)


@pytest.mark.parametrize("ishape,oshape", [((77, 123), (64, 96)), ((300, 500), (64, 96)), ((31, 45), (62, 90)),
                                           ((64, 96), (64, 96)), ((540, 1024), (96, 160))])
def test_host_coefficient_tables_reproduce_pillow(ishape, oshape):
    """CPU check of glsdet_amd.preprocess.pil_bicubic_tables: fixed-point two-pass resampling with
    these tables == PIL.Image.resize(BICUBIC) on uint8, bit for bit."""
    from PIL import Image
    img = synth_image(ishape, 3)
    want = np.array(Image.fromarray(img, "RGB").resize((oshape[1], oshape[0]), Image.BICUBIC))
    assert np.array_equal(_two_pass_numpy(img, oshape), want)


@pytest.mark.gpu
@pytest.mark.parametrize("i", range(len(CASES)))
def test_hip_preprocess_equals_the_reference_bit_for_bit(pre_golden, i):
    from glsdet_amd.preprocess import DronePreprocessor
    ishape, shape, lb = CASES[i]
    out = DronePreprocessor()([synth_image(ishape, i)], shape, lb)
    assert out.dtype == torch.float32 and tuple(out.shape) == (1, 3) + tuple(shape)
    assert np.array_equal(out.cpu().numpy(), pre_golden["pre/%d" % i])


@pytest.mark.gpu
@pytest.mark.parametrize("lb", [False, True])
def test_hip_preprocess_batch_of_mixed_sizes_vs_oracle(lb):
    """UAVDT-sized (540x1024) and other frames into one 8x3x640x640 batch (the reference's input)."""
    from glsdet_amd.preprocess import DronePreprocessor
    sizes = [(540, 1024), (427, 640), (333, 517), (640, 640), (1080, 1920), (200, 150), (540, 1024), (64, 48)]
    imgs = [synth_image(s, 10 + j) for j, s in enumerate(sizes)]
    out = DronePreprocessor()(imgs, (640, 640), lb).cpu().numpy()
    for j, im in enumerate(imgs):
        assert np.array_equal(out[j:j + 1], P.drone_preprocess(im, (640, 640), lb)), "image %d" % j


@pytest.mark.gpu
def test_hip_preprocess_argument_errors():
    from glsdet_amd.preprocess import DronePreprocessor
    p = DronePreprocessor()
    with pytest.raises(ValueError):
        p([np.zeros((8, 8, 3), np.float32)], (64, 64))
    with pytest.raises(ValueError):
        p([np.zeros((8, 8), np.uint8)], (64, 64))
