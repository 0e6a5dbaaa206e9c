"""Deterministic synthetic weights / inputs (input DATA for tests and bench; no arithmetic
of the detection path lives here).  Shared by the golden generator, the oracle tests, the
GPU tests and bench.py so that a state_dict never has to be stored."""
import math
from typing import Dict, Tuple

import numpy as np
import torch


def synth_tensor(key: str, shape: Tuple[int, ...], seed: int) -> np.ndarray:
    """Deterministic parameter filler shared by the golden generator, the oracle tests and
    the product tests/bench: every tensor depends only on (key, shape, seed), so a
    state_dict never has to be stored.  BN stats/affine and every bias are perturbed
    (default init yields near-constant logits)."""
    import zlib
    rng = np.random.default_rng([seed, zlib.crc32(key.encode())])
    leaf = key.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return np.zeros(shape, np.int64)
    if leaf == "scale":                                  # mmcv Scale (gfl_head.py:151)
        return rng.uniform(0.7, 1.3, shape).astype(np.float32)
    if leaf in ("proxies", "_embedding"):               # mp_head.py:78-91
        return rng.standard_normal(shape).astype(np.float32)
    if leaf == "project":                               # Integral buffer, gfl_head.py:32-33
        return np.linspace(0, shape[0] - 1, shape[0]).astype(np.float32)
    if leaf == "_pos_embedding_ptr":
        return np.zeros(shape, np.int64)
    if leaf == "_proxies_prob":
        return rng.uniform(0.1, 0.5, shape).astype(np.float32)
    if leaf == "running_var":
        return rng.uniform(0.5, 1.5, shape).astype(np.float32)
    if leaf == "running_mean":
        return (0.1 * rng.standard_normal(shape)).astype(np.float32)
    if leaf == "weight" and len(shape) == 1:            # BN gamma
        return rng.uniform(0.7, 1.3, shape).astype(np.float32)
    if leaf == "bias":
        return (0.2 * rng.standard_normal(shape)).astype(np.float32)
    if leaf == "weight":
        fan_in = int(np.prod(shape[1:]))
        w = rng.standard_normal(shape) * math.sqrt(1.0 / fan_in)
        if len(shape) == 4 and fan_in > 1:
            # zero-mean filters: the (always positive) mean of SiLU activations is not
            # passed on, which keeps random-weight nets from amplifying a DC component
            w = w - w.mean(axis=(1, 2, 3), keepdims=True)
        return w.astype(np.float32)
    raise KeyError(key)


def synth_state_dict(shapes: Dict[str, Tuple[int, ...]], seed: int) -> Dict[str, torch.Tensor]:
    return {k: torch.from_numpy(synth_tensor(k, tuple(s), seed)) for k, s in shapes.items()}


def synth_input(shape: Tuple[int, ...], seed: int) -> torch.Tensor:
    rng = np.random.default_rng([seed, 0x1234])
    return torch.from_numpy(rng.standard_normal(shape).astype(np.float32))


def synth_resdet_state_dict(kind: str, seed: int, x: torch.Tensor, **kw) -> Dict[str, torch.Tensor]:
    """Synthetic ResNet + FPN + GFL/MP state_dict whose BN running statistics are the
    (perturbed) batch statistics of input x, computed block by block with plain torch ops in
    the order the backbone executes -- the recipe of tests/golden/make_golden.calibrate_bn, so
    the signal neither dies nor explodes over the 53 conv layers.  Data preparation only."""
    import torch.nn.functional as F
    from .arch import RESNET_STAGE_BLOCKS, resdet_state_dict_shapes
    sd = synth_state_dict(resdet_state_dict_shapes(kind, **kw), seed)
    rng = np.random.default_rng([seed, 0xB17])

    def bn(p, t):
        var = t.var((0, 2, 3), unbiased=False)
        var = var + 0.5 * var.mean() + 1e-4
        mean = t.mean((0, 2, 3))
        c = mean.numel()
        sd[p + ".running_var"] = var * torch.from_numpy(rng.uniform(0.8, 1.25, c).astype(np.float32))
        sd[p + ".running_mean"] = mean + var.sqrt() * torch.from_numpy((0.1 * rng.standard_normal(c)).astype(np.float32))
        return F.batch_norm(t, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"],
                            False, 0.0, 1e-5)

    with torch.no_grad():
        t = torch.relu(bn("backbone.bn1", F.conv2d(x, sd["backbone.conv1.weight"], None, 2, 3)))
        t = F.max_pool2d(t, 3, 2, 1)
        for i, nb in enumerate(RESNET_STAGE_BLOCKS[kw.get("depth", 50)]):
            for j in range(nb):
                p = "backbone.layer%d.%d" % (i + 1, j)
                s = 2 if (j == 0 and i > 0) else 1
                o = torch.relu(bn(p + ".bn1", F.conv2d(t, sd[p + ".conv1.weight"])))
                o = torch.relu(bn(p + ".bn2", F.conv2d(o, sd[p + ".conv2.weight"], None, s, 1)))
                o = bn(p + ".bn3", F.conv2d(o, sd[p + ".conv3.weight"]))
                idn = t
                if p + ".downsample.0.weight" in sd:
                    idn = bn(p + ".downsample.1", F.conv2d(t, sd[p + ".downsample.0.weight"], None, s))
                t = torch.relu(o + idn)
    return sd
