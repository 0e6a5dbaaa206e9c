"""Shared test helpers: rebuild the state_dicts / inputs the golden generator used."""
import json

import numpy as np
import torch

from oracle import glsdet_oracle as O


def meta_of(golden, key):
    return json.loads(bytes(golden[key]).decode())


def block_case(golden, tag):
    """-> (state_dict with 'm.' prefix, input, expected output)"""
    meta = meta_of(golden, "block/%s/meta" % tag)
    sd = {"m." + k: torch.from_numpy(O.synth_tensor(k, tuple(s), meta["seed"]))
          for k, s in meta["shapes"].items()}
    pre = "block/%s/bn/" % tag                 # calibrated BN statistics, where the case stores them
    for k in golden.files:
        if k.startswith(pre):
            sd["m." + k[len(pre):]] = torch.from_numpy(golden[k])
    x = O.synth_input(tuple(meta["in_shape"]), meta["seed"] + 100)
    return sd, x, torch.from_numpy(golden["block/%s/y" % tag])


def model_tags(golden):
    return sorted({k.split("/")[1] for k in golden.files if k.startswith("model/")})


def model_case(golden, shapes, tag):
    """tag like 'gl_tiny_seed0' -> (meta, state_dict, input, [expected outs], decoded)"""
    meta = meta_of(golden, "model/%s/meta" % tag)
    sh = shapes["%s_%s" % (meta["model"], meta["phi"])]
    sd = O.synth_state_dict(sh, meta["seed"])
    pre = "model/%s/bn/" % tag
    for k in golden.files:
        if k.startswith(pre):
            sd[k[len(pre):]] = torch.from_numpy(golden[k])
    x = O.synth_input(tuple(meta["in_shape"]), meta["seed"] + 100)
    outs = [torch.from_numpy(golden["model/%s/out%d" % (tag, i)]) for i in range(3)]
    return meta, sd, x, outs, torch.from_numpy(golden["model/%s/decoded" % tag])


def calibrated_resdet_sd(kind, seed, x, **kw):
    """Synthetic ResNet-50 + FPN + GFL/MP state_dict whose BN running statistics are the
    (perturbed) batch statistics of input x, computed stage by stage with plain torch ops in
    the order the backbone executes -- the same recipe as make_golden.calibrate_bn, so the
    signal neither dies nor explodes over the 53 conv layers.  Data preparation only."""
    import numpy as np
    import torch.nn.functional as F
    from glsdet_amd.arch import RESNET_STAGE_BLOCKS, resdet_state_dict_shapes
    sd = O.synth_state_dict(resdet_state_dict_shapes(kind, **kw), seed)
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
