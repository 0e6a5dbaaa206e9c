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
    """Synthetic ResNet-50 + FPN + GFL/MP weights with BN statistics calibrated on x (data only)."""
    from glsdet_amd.synth import synth_resdet_state_dict
    return synth_resdet_state_dict(kind, seed, x, **kw)
