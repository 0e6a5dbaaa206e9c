#!/usr/bin/env python3
"""Golden-vector generator -- runs ONLY in the build container, where the reference is
mounted read-only at /root/reference.  It imports the reference's own CPU implementation
(yolox-drone tree), fills every parameter/buffer with the deterministic filler
``oracle.glsdet_oracle.synth_tensor`` (so no state_dict has to be stored), runs the
reference forward and writes inputs' seeds + expected outputs to ``tests/golden/*.npz``
and the reference's state_dict key->shape tables to ``tests/golden/shapes_*.json``.

The reference never travels to the GPU box; these small data files do.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/yolox-drone")
sys.dont_write_bytecode = True

from oracle.glsdet_oracle import synth_input, synth_tensor  # noqa: E402

torch.set_grad_enabled(False)
torch.manual_seed(0)


def fill(module, seed):
    sd = module.state_dict()
    new = {k: torch.from_numpy(synth_tensor(k, tuple(v.shape), seed)) for k, v in sd.items()}
    module.load_state_dict(new)
    module.eval()
    return {k: list(v.shape) for k, v in sd.items()}


def calibrate_bn(net, x, seed):
    """Whole models only: replace the filler's BN running stats by perturbed batch
    statistics of a synthetic input, so the signal neither dies nor explodes over ~80
    layers and the logits stay input-sensitive.  Done in ONE eval-mode forward with a
    pre-hook per BN (execution order = topological order), so every BN is calibrated on
    what its already-perturbed upstream really produces.
    Returns {key: ndarray} for every running_mean / running_var (stored in the fixture:
    these are the only tensors that are NOT a pure function of (key, shape, seed))."""
    rng = np.random.default_rng([seed, 0xB17])

    def pre(m, inp):
        t = inp[0]
        var = t.var((0, 2, 3), unbiased=False)
        var = var + 0.5 * var.mean() + 1e-4      # floor: no channel may amplify by > ~1.4x
        mean = t.mean((0, 2, 3))
        c = mean.numel()
        m.running_var.copy_(var * torch.from_numpy(rng.uniform(0.8, 1.25, c).astype(np.float32)))
        m.running_mean.copy_(mean + var.sqrt() * torch.from_numpy(
            (0.1 * rng.standard_normal(c)).astype(np.float32)))

    hooks = [m.register_forward_pre_hook(pre) for m in net.modules()
             if isinstance(m, torch.nn.BatchNorm2d)]
    net.eval()
    net(x)
    for h in hooks:
        h.remove()
    return {k: v.numpy().copy() for k, v in net.state_dict().items()
            if k.endswith("running_mean") or k.endswith("running_var")}


def make_block(out):
    """-> block(tag, ctor, in_shape, seed=0, calibrate=False): run a reference module on seeded data into `out`."""
    def block(tag, ctor, in_shape, seed=0, calibrate=False):
        m = ctor()
        shapes = fill(m, seed)
        x = synth_input(in_shape, seed + 100)
        if calibrate:
            for k, v in calibrate_bn(m, x, seed).items():
                out["block/%s/bn/%s" % (tag, k)] = v
        y = m(x)
        out["block/%s/y" % tag] = y.numpy()
        out["block/%s/meta" % tag] = np.frombuffer(json.dumps(
            {"shapes": shapes, "in_shape": list(in_shape), "seed": seed}).encode(), np.uint8)
    return block


def attention_cases(block):
    """SURVEY section 8a row A6': the other GL attention variants the tree holds
    (drone/models/new/Non_local_family.py, new/darknet_att.py)."""
    from models.new.Non_local_family import Attention, Patch_Conv_NonLocal_new, SpatialAttention
    from models.new.darknet_att import CSPDarknet as AttDarknet
    block("att_pcnl_new_nonlinear", lambda: Patch_Conv_NonLocal_new(32, 32, channel_scale=1), (2, 32, 20, 24))
    block("att_pcnl_new_linear", lambda: Patch_Conv_NonLocal_new(32, 48, channel_scale=1, channel_cat="linear"),
          (2, 32, 21, 27))
    block("att_attention_c32", lambda: Attention(32), (2, 32, 20, 24))
    block("att_attention_c48_odd", lambda: Attention(48), (1, 48, 17, 23))
    block("att_spatial_attention", lambda: SpatialAttention(7), (2, 32, 20, 24))
    from models.new.Non_local_family import Patch_Conv_NonLocal_44
    block("att_pcnl_44", lambda: Patch_Conv_NonLocal_44(32, 64, channel_scale=0.5), (2, 32, 40, 48))
    block("att_pcnl_44_odd", lambda: Patch_Conv_NonLocal_44(16, 32, channel_scale=0.5), (1, 16, 36, 44))
    from models.new.Non_local_family import Patch_Conv_NonLocal_adapt, Patch_Conv_NonLocal_adapt_new
    block("att_pcnl_adapt", lambda: Patch_Conv_NonLocal_adapt(32, 64, channel_scale=1), (2, 32, 24, 36))
    block("att_pcnl_adapt_nonlinear", lambda: Patch_Conv_NonLocal_adapt(16, 32, channel_scale=1, channel_cat="non_linear"),
          (3, 16, 32, 28), seed=1)
    block("att_pcnl_adapt_new", lambda: Patch_Conv_NonLocal_adapt_new(32, 64, channel_scale=0.5), (2, 32, 24, 36))
    block("att_pcnl_adapt_new_linear", lambda: Patch_Conv_NonLocal_adapt_new(16, 16, channel_scale=0.5, channel_cat="linear"),
          (3, 16, 30, 22), seed=1)

    # the LSK gating unit (drone/models/lsk/LSK.py; darknet_lsk.py = darknet_att.py with this Attention)
    from models.lsk.LSK import Attention as LskAttention, LSKblock
    from models.lsk.darknet_lsk import CSPDarknet as LskDarknet
    block("att_lskblock_c32", lambda: LSKblock(32), (2, 32, 20, 24))
    block("att_lsk_attention_c48_odd", lambda: LskAttention(48), (1, 48, 17, 23), seed=1)

    def outs_of(ctor):
        class Outs(torch.nn.Module):            # dict -> tuple so that block() can store it
            def __init__(self):
                super().__init__()
                self.backbone = ctor(0.33, 0.375, out_features=("dark2", "dark3", "dark4", "dark5"))

            def forward(self, x):
                f = self.backbone(x)
                return torch.cat([f[k].flatten(1) for k in ("dark2", "dark3", "dark4", "dark5")], 1)
        return Outs
    block("att_darknet_tiny", outs_of(AttDarknet), (1, 3, 128, 160), calibrate=True)
    block("att_lsk_darknet_tiny", outs_of(LskDarknet), (1, 3, 128, 160), calibrate=True)


PRE_CASES = [  # (in_h, in_w), input_shape (H, W), letterbox   -- synth_image(seed = index)
    ((77, 123), (64, 96), False), ((77, 123), (64, 96), True), ((150, 90), (64, 96), True),
    ((64, 96), (64, 96), False), ((300, 500), (64, 96), False), ((31, 45), (64, 96), True),
    ((97, 64), (96, 64), True),
]


def synth_image(shape, seed):
    """Smooth colour gradients + texture + saturated patches, uint8 HWC RGB (test input data)."""
    h, w = shape
    rng = np.random.default_rng([seed, 0x1A6E])
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    img = np.stack([127 + 120 * np.sin(xx / (5 + 3 * c) + seed) * np.cos(yy / (7 + 2 * c)) for c in range(3)], -1)
    img += rng.normal(0, 25, img.shape)
    img[: h // 5, : w // 4] = 255
    img[-(h // 6):, -(w // 5):] = 0
    return np.clip(img, 0, 255).astype(np.uint8)


def preprocess_cases():
    """Row f.3: the reference's own resize_image + preprocess_input (drone/models/core/utils.py)."""
    from PIL import Image
    from models.core.utils import preprocess_input, resize_image
    out = {}
    for i, (ishape, shape, lb) in enumerate(PRE_CASES):
        img = Image.fromarray(synth_image(ishape, i), "RGB")
        data = resize_image(img, (shape[1], shape[0]), lb)
        out["pre/%d" % i] = np.expand_dims(np.transpose(preprocess_input(np.array(data, dtype="float32")), (2, 0, 1)), 0)
    np.savez_compressed(os.path.join(HERE, "preprocess_golden.npz"), **out)
    print("preprocess:", len(out), "cases,", os.path.getsize(os.path.join(HERE, "preprocess_golden.npz")), "bytes")


def ufp_boxes(trial):
    """Seeded coarse-detection-like xyxy boxes for one image (input data of the packing goldens)."""
    rng = np.random.default_rng([trial, 0x0F9])
    n = int(rng.integers(1, 60))
    W, H = int(rng.integers(400, 2000)), int(rng.integers(300, 1200))
    c = rng.uniform(0, 1, (n, 2)) * [W, H]
    wh = np.exp(rng.uniform(np.log(6), np.log(220), (n, 2)))
    b = np.concatenate([c - wh / 2, c + wh / 2], 1)
    b[:, 0::2] = np.clip(b[:, 0::2], 0, W - 1)
    b[:, 1::2] = np.clip(b[:, 1::2], 0, H - 1)
    return (b.astype(np.float32) if trial % 3 == 0 else b), W, H


def ufp_cases():
    """Rows f.1: the reference's UnifiedForegroundPacking (ufp/UFPMP-Det-Tools/ufp, pure numpy)."""
    sys.path.insert(0, "/root/reference/yolox-ufp/UFPMP-Det-Tools")
    from ufp import UnifiedForegroundPacking
    out = {}
    for trial in range(24):
        b, W, H = ufp_boxes(trial)
        chips, cw, ch = UnifiedForegroundPacking(b.copy(), 1.5, input_shape=[W, H])
        out["ufp/%d/chips" % trial] = np.asarray(chips, np.float64).reshape(-1, 7)
        out["ufp/%d/canvas" % trial] = np.asarray([cw, ch], np.float64)
    np.savez_compressed(os.path.join(HERE, "ufp_golden.npz"), **out)
    print("ufp:", len(out) // 2, "cases,", os.path.getsize(os.path.join(HERE, "ufp_golden.npz")), "bytes")


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def merge_cases():
    """Row f.2: the reference's own pure-numpy `compute_iof` and `py_cpu_nms` (ufp/ufpmp_det_eval.py:36-50, 149-178).
    The script imports mmdet / mmcv / cv2 / pycocotools at its top; none of them is installed and none is touched by
    the two functions, so empty stand-in modules satisfy the imports (as for torchvision above) and the functions
    themselves run as shipped.  Inputs: seeded boxes; outputs: IoF values and keep lists."""
    none = lambda *a, **k: None
    _stub("mmdet")
    _stub("mmdet.apis", init_detector=none, show_result_pyplot=none, inference_detector=none)
    _stub("cv2")
    _stub("mmcv")
    _stub("mmcv.parallel", collate=none, scatter=none)
    _stub("mmdet.datasets")
    _stub("mmdet.datasets.pipelines", Compose=none)
    _stub("mmdet.core", UnifiedForegroundPacking=none)
    _stub("pycocotools")
    _stub("pycocotools.coco", COCO=none)
    _stub("pycocotools.cocoeval", COCOeval=none)
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_ufpmp_det_eval", "/root/reference/yolox-ufp/ufpmp_det_eval.py")
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    for k in ("mmdet", "mmdet.apis", "cv2", "mmcv", "mmcv.parallel", "mmdet.datasets", "mmdet.datasets.pipelines", "mmdet.core",
              "pycocotools", "pycocotools.coco", "pycocotools.cocoeval"):
        sys.modules.pop(k, None)
    out = {}
    rng = np.random.default_rng(0x10F)
    # compute_iof: box pairs incl. disjoint, touching, contained, degenerate-but-positive
    a = rng.uniform(0, 200, (400, 2))
    b = rng.uniform(0, 200, (400, 2))
    A = np.concatenate([a, a + rng.uniform(1, 120, (400, 2))], 1)
    B = np.concatenate([b, b + rng.uniform(1, 120, (400, 2))], 1)
    B[:40] = A[:40]                                            # identical
    B[40:80, :2] = A[40:80, 2:]                                # touching at a corner -> 0
    B[80:120] = np.concatenate([A[80:120, :2] + 1, A[80:120, :2] + 2], 1)   # small box inside (or outside) the other
    A32 = A.astype(np.float32)
    out["iof/a"], out["iof/b"] = A, B
    out["iof/value"] = np.array([ref.compute_iof(list(x), list(y)) for x, y in zip(A, B)], np.float64)
    out["iof/value_f32_first"] = np.array([ref.compute_iof(list(x), list(y)) for x, y in zip(A32, B)], np.float64)
    # py_cpu_nms: clustered boxes, several thresholds, with and without score ties
    for case in range(12):
        r = np.random.default_rng([case, 0x2C5])
        n = int(r.integers(1, 90)) if case else 1
        centres = r.uniform(120, 400, (max(1, n // 6), 2))        # boxes stay at positive coordinates (wh <= 90, jitter 6 sigma)
        c = centres[r.integers(0, len(centres), n)] + r.normal(0, 6, (n, 2))
        wh = r.uniform(8, 90, (n, 2))
        sc = r.uniform(0.05, 1.0, n)
        if case % 3 == 2:
            sc = np.round(sc, 1)                               # ties: the reference order is numpy's argsort()[::-1]
        dets = np.concatenate([c - wh / 2, c + wh / 2, sc[:, None]], 1)
        if case % 4 == 1:
            dets = dets.astype(np.float32)
        thr = [0.6, 0.5, 0.3, 0.65][case % 4]
        out["nms/%d/dets" % case] = dets
        out["nms/%d/thr" % case] = np.float64(thr)
        out["nms/%d/keep" % case] = np.asarray(ref.py_cpu_nms(dets, thr), np.int64)
    np.savez_compressed(os.path.join(HERE, "merge_golden.npz"), **out)
    print("merge:", len(out), "arrays,", os.path.getsize(os.path.join(HERE, "merge_golden.npz")), "bytes")


class _CocoBox:
    """Input container for the reference's COCOeval: the four accessors it calls, with pycocotools.COCO's
    documented ordering (getAnnIds: by image in the order given, annotation order inside an image, then the
    category filter).  Data plumbing only."""

    def __init__(self, dataset):
        self.dataset = dataset
        self.anns = {a["id"]: a for a in dataset["annotations"]}
        self.by_img = {}
        for a in dataset["annotations"]:
            self.by_img.setdefault(a["image_id"], []).append(a)

    def getImgIds(self):
        return [im["id"] for im in self.dataset["images"]]

    def getCatIds(self):
        return [c["id"] for c in self.dataset["categories"]]

    def getAnnIds(self, imgIds=[], catIds=[]):
        anns = [a for i in imgIds for a in self.by_img.get(i, [])] if len(imgIds) else list(self.dataset["annotations"])
        if len(catIds):
            anns = [a for a in anns if a["category_id"] in catIds]
        return [a["id"] for a in anns]

    def loadAnns(self, ids):
        return [self.anns[i] for i in ids]


def eval_cases():
    """Row f.4: the reference's vendored COCOeval (drone/models/core/cocoeval.py) run on seeded data sets:
    `_prepare`, `computeIoU` (sorting, maxDets cut), `evaluateImg`, `accumulate`, `summarize` are the reference's
    own code.  The ONE piece that is not: `maskUtils.iou`, compiled code of pycocotools (`_mask`, absent) -- the
    stand-in module routes it to the restatement `cocoeval_oracle.bb_iou` (maskApi.c bbIou), which therefore stays
    unpinned.  `np.float` (removed from numpy 1.24; the file predates that) is restored as the alias of `float` it
    always was."""
    import copy
    from oracle import cocoeval_oracle as CO
    from tests.test_cocoeval import random_case

    def iou(d, g, iscrowd):
        if len(d) == 0 or len(g) == 0:
            return []
        return CO.bb_iou(d, g, iscrowd)
    _stub("pycocotools")
    _stub("pycocotools._mask", iou=iou, merge=None, frPyObjects=None)      # mask.py:76-78 binds the three names at import
    if not hasattr(np, "float"):
        np.float = float
    from models.core.cocoeval import COCOeval
    out = {}
    cases = [(0, {}, (1, 10, 25), "coco", 1), (1, dict(zero_id=True), (1, 10, 25), "coco", 1), (2, dict(crowd=0.5), (1, 10, 25), "coco", 0),
             (3, dict(n_img=1, n_cat=1, max_gt=70, max_dt=150), (10, 100, 120), "coco", 1),
             (5, dict(big=True), (1, 10, 25), "drone", 1), (6, dict(n_img=12, n_cat=10), (10, 100, 500), "coco", 1),
             (7, dict(crowd=0.0, ties=False), (1, 10, 25), "coco", 0)]
    meta = []
    for ci, (seed, kw, max_dets, area, use_cats) in enumerate(cases):
        ds, res = random_case(100 + seed, **kw)
        gt = _CocoBox(copy.deepcopy(ds))
        dt = _CocoBox(CO.load_res(ds, res))
        E = COCOeval(gt, dt, "bbox")
        E.params.imgIds = sorted(gt.getImgIds())
        E.params.catIds = sorted(gt.getCatIds())
        E.params.maxDets = list(max_dets)
        E.params.areaRng = [list(r) for r in (CO.DRONE_AREA if area == "drone" else CO.COCO_AREA)]
        E.params.useCats = use_cats
        E.evaluate()
        E.accumulate()
        E.summarize()
        pre = "eval/%d/" % ci
        out[pre + "stats"] = np.asarray(E.stats, np.float64)
        for k in ("precision", "recall", "scores"):
            out[pre + k] = np.asarray(E.eval[k], np.float64)
        out[pre + "none"] = np.asarray([e is None for e in E.evalImgs], np.uint8)
        for i, e in enumerate(E.evalImgs):
            if e is None:
                continue
            out[pre + "img%d/dtm" % i] = np.asarray(e["dtMatches"], np.float64)
            out[pre + "img%d/gtm" % i] = np.asarray(e["gtMatches"], np.float64)
            out[pre + "img%d/dtig" % i] = np.asarray(e["dtIgnore"], np.uint8)
            out[pre + "img%d/gtig" % i] = np.asarray(e["gtIgnore"], np.uint8)
            out[pre + "img%d/ids" % i] = np.asarray(list(e["dtIds"]) + [-1] + list(e["gtIds"]), np.int64)
        meta.append(dict(seed=100 + seed, kw=kw, max_dets=list(max_dets), area=area, use_cats=use_cats,
                         n_eval=len(E.evalImgs)))
    out["eval/meta"] = np.frombuffer(json.dumps(meta).encode(), np.uint8)
    sys.modules.pop("pycocotools", None)
    sys.modules.pop("pycocotools._mask", None)
    np.savez_compressed(os.path.join(HERE, "eval_golden.npz"), **out)
    print("eval:", len(cases), "cases,", os.path.getsize(os.path.join(HERE, "eval_golden.npz")), "bytes")


def _mmcv_building_blocks():
    """Stand-ins for the FIVE mmcv names resnet.py / res_layer.py / fpn.py import (mmcv-full is not installable here).
    Each is the documented behaviour of the mmcv function for the arguments these files pass -- nothing of the reference's
    own composition logic (strides, 'pytorch' style, downsample, stage structure, top-down path, extra convs) is restated:
      build_conv_layer(None, ...)          -> nn.Conv2d(...)
      build_norm_layer(dict(type='BN'), c, postfix) -> ('bn<postfix>', nn.BatchNorm2d(c))    (requires_grad honoured)
      build_plugin_layer                   -> never called (plugins=None)
      BaseModule / Sequential              -> nn.Module / nn.Sequential accepting init_cfg
      ConvModule(conv_cfg=None, norm_cfg=None, act_cfg=None) -> .conv = nn.Conv2d(bias=True); forward = conv
      auto_fp16 / force_fp32               -> identity decorators
    """
    mmcv = _stub("mmcv")
    cnn = _stub("mmcv.cnn")
    runner = _stub("mmcv.runner")

    def build_conv_layer(cfg, *a, **k):
        assert cfg is None
        return torch.nn.Conv2d(*a, **k)

    def build_norm_layer(cfg, num_features, postfix=""):
        assert cfg["type"] == "BN"
        m = torch.nn.BatchNorm2d(num_features, eps=cfg.get("eps", 1e-5))
        for p_ in m.parameters():
            p_.requires_grad = cfg.get("requires_grad", True)
        return "bn" + str(postfix), m

    class BaseModule(torch.nn.Module):
        def __init__(self, init_cfg=None):
            super().__init__()
            self.init_cfg = init_cfg

    class Sequential(torch.nn.Sequential):
        def __init__(self, *a, init_cfg=None):
            super().__init__(*a)

    class ConvModule(torch.nn.Module):
        def __init__(self, cin, cout, k, stride=1, padding=0, conv_cfg=None, norm_cfg=None, act_cfg=dict(type="ReLU"), inplace=True):
            super().__init__()
            assert conv_cfg is None and norm_cfg is None and act_cfg is None, "FPN builds plain convs"
            self.conv = torch.nn.Conv2d(cin, cout, k, stride, padding)

        def forward(self, x):
            return self.conv(x)

    ident = lambda *a, **k: (lambda f: f)
    cnn.build_conv_layer, cnn.build_norm_layer, cnn.build_plugin_layer, cnn.ConvModule = build_conv_layer, build_norm_layer, None, ConvModule
    runner.BaseModule, runner.Sequential, runner.auto_fp16, runner.force_fp32 = BaseModule, Sequential, ident, ident
    mmcv.cnn, mmcv.runner = cnn, runner
    return mmcv


def _load_ref_module(name, path):
    import importlib.util
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def resdet_cases():
    """the reference's OWN mmdet/models/backbones/resnet.py (ResNet, Bottleneck),
    utils/res_layer.py (ResLayer) and necks/fpn.py (FPN) executed on seeded data -- loaded by file path under their package
    names with the mmcv building blocks stood in (above) and an empty registry.  Pins the composition the restatement
    (oracle/mpdet_oracle.py) and the HIP path (glsdet_amd/resdet.py) must reproduce."""
    saved = {k: sys.modules.get(k) for k in ("mmcv", "mmcv.cnn", "mmcv.runner", "mmdet", "mmdet.models", "mmdet.models.builder",
                                             "mmdet.models.utils", "mmdet.models.utils.res_layer", "mmdet.models.backbones",
                                             "mmdet.models.backbones.resnet", "mmdet.models.necks", "mmdet.models.necks.fpn")}
    try:
        _mmcv_building_blocks()
        root = os.path.join("/root/reference", "yolox-ufp", "mmdet", "models")
        for pkg in ("mmdet", "mmdet.models", "mmdet.models.utils", "mmdet.models.backbones", "mmdet.models.necks"):
            m = _stub(pkg)
            m.__path__ = []
        reg = type("Registry", (), {"register_module": lambda self, *a, **k: (lambda c: c)})()
        b = _stub("mmdet.models.builder")
        b.BACKBONES = b.NECKS = reg
        rl = _load_ref_module("mmdet.models.utils.res_layer", os.path.join(root, "utils", "res_layer.py"))
        sys.modules["mmdet.models.utils"].ResLayer = rl.ResLayer
        rn = _load_ref_module("mmdet.models.backbones.resnet", os.path.join(root, "backbones", "resnet.py"))
        fp = _load_ref_module("mmdet.models.necks.fpn", os.path.join(root, "necks", "fpn.py"))
        out = {}
        block = make_block(out)

        class Trunk(torch.nn.Module):       # configs/UFPMP-Det: ResNet-50 (4 stages, out 0..3, BN eval, pytorch style) + FPN
            def __init__(self, fpn_kw):
                super().__init__()
                self.backbone = rn.ResNet(depth=50, num_stages=4, out_indices=(0, 1, 2, 3), frozen_stages=1,
                                          norm_cfg=dict(type="BN", requires_grad=True), norm_eval=True, style="pytorch")
                self.neck = fp.FPN(in_channels=[256, 512, 1024, 2048], out_channels=256, **fpn_kw) if fpn_kw is not None else None

            def forward(self, x):
                f = self.backbone(x)
                if self.neck is not None:
                    f = self.neck(f)
                return torch.cat([t.flatten(1) for t in f], 1)
        with torch.no_grad():
            block("res50_c2_c5", lambda: Trunk(None), (1, 3, 64, 96), calibrate=True)
            block("res50_fpn_gfl", lambda: Trunk(dict(start_level=1, add_extra_convs="on_output", num_outs=5)), (2, 3, 96, 128),
                  seed=1, calibrate=True)
            block("res50_fpn_all_levels_odd", lambda: Trunk(dict(start_level=0, add_extra_convs="on_input", num_outs=5)), (1, 3, 72, 104),
                  seed=2, calibrate=True)
            block("res_bottleneck_s2_down", lambda: rn.Bottleneck(64, 32, stride=2, style="pytorch",
                                                                  downsample=torch.nn.Sequential(torch.nn.Conv2d(64, 128, 1, 2, bias=False),
                                                                                                 torch.nn.BatchNorm2d(128))),
                  (2, 64, 20, 24), seed=3)
        np.savez_compressed(os.path.join(HERE, "resdet_golden.npz"), **out)
        print("resdet:", len(out), "arrays,", os.path.getsize(os.path.join(HERE, "resdet_golden.npz")), "bytes")
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def head_cases():
    """SURVEY section 8a row A11 (forward part): the reference's OWN GFLHead / MPHead / Integral
    (yolox-ufp/mmdet/models/dense_heads/{gfl_head,mp_head,anchor_head,base_dense_head,dense_test_mixins}.py), loaded by file
    path.  Stand-ins: the mmcv building blocks (ConvModule = conv without bias -> GroupNorm -> ReLU with mmcv's attribute
    names `conv`, `gn`; Scale = x * scalar parameter; init helpers = no-ops), `mmdet.core` as a namespace of dummies except
    multi_apply (mmcv's five-line map/zip), nltk (imported by mp_head.py for training only), an empty registry, and
    constructor-only dummies for build_loss / build_bbox_coder / build_prior_generator (strides and one prior per position are
    all __init__ reads).  What is pinned: the towers, predictors, per-level Scale, the fp32 cast, MPHead's gfl_cls_conv +
    forward_proxy (normalisation, per-class softmax-weighted mean over its proxies, gamma), Integral."""
    names = ["mmcv", "mmcv.cnn", "mmcv.cnn.utils", "mmcv.cnn.utils.weight_init", "mmcv.runner", "mmcv.ops", "nltk", "nltk.cluster",
             "nltk.cluster.kmeans", "mmdet", "mmdet.core", "mmdet.core.utils", "mmdet.utils", "mmdet.utils.contextmanagers", "mmdet.models",
             "mmdet.models.builder",
             "mmdet.models.dense_heads", "mmdet.models.dense_heads.base_dense_head", "mmdet.models.dense_heads.dense_test_mixins",
             "mmdet.models.dense_heads.anchor_head", "mmdet.models.dense_heads.gfl_head", "mmdet.models.dense_heads.mp_head",
             "mmdet.core.anchor", "mmdet.core.anchor.builder", "mmdet.core.anchor.anchor_generator", "mmdet.core.bbox",
             "mmdet.core.bbox.builder", "mmdet.core.bbox.transforms", "mmdet.core.bbox.coder", "mmdet.core.bbox.coder.base_bbox_coder",
             "mmdet.core.bbox.coder.distance_point_bbox_coder", "mmdet.core.mask", "mmdet.core.mask.structures", "mmdet.core.utils.misc"]
    saved = {k: sys.modules.get(k) for k in names}
    try:
        mmcv = _mmcv_building_blocks()
        nn = torch.nn

        class ConvModule(nn.Module):
            def __init__(self, cin, cout, k, stride=1, padding=0, conv_cfg=None, norm_cfg=None, act_cfg=dict(type="ReLU"), inplace=True):
                super().__init__()
                assert conv_cfg is None and norm_cfg is not None and norm_cfg["type"] == "GN" and act_cfg["type"] == "ReLU"
                self.conv = nn.Conv2d(cin, cout, k, stride, padding, bias=False)      # bias='auto': none under a norm
                self.gn = nn.GroupNorm(norm_cfg["num_groups"], cout)
                self.activate = nn.ReLU(inplace=inplace)

            def forward(self, x):
                return self.activate(self.gn(self.conv(x)))

        class Scale(nn.Module):
            def __init__(self, scale=1.0):
                super().__init__()
                self.scale = nn.Parameter(torch.tensor(scale, dtype=torch.float))

            def forward(self, x):
                return x * self.scale
        noop = lambda *a, **k: None
        mmcv.cnn.ConvModule, mmcv.cnn.Scale, mmcv.cnn.bias_init_with_prob, mmcv.cnn.normal_init = ConvModule, Scale, (lambda p_: 0.0), noop
        u = _stub("mmcv.cnn.utils")
        wi = _stub("mmcv.cnn.utils.weight_init", constant_init=noop)
        u.weight_init = wi
        _stub("mmcv.ops", batched_nms=None)
        _stub("nltk")
        _stub("nltk.cluster")
        _stub("nltk.cluster.kmeans", KMeansClusterer=None)

        def multi_apply(func, *args, **kwargs):           # mmdet/core/utils/misc.py:10-30
            from functools import partial
            pfunc = partial(func, **kwargs) if kwargs else func
            return tuple(map(list, zip(*map(pfunc, *args))))

        class _Prior:
            strides = [(8, 8), (16, 16), (32, 32), (64, 64), (128, 128)]
            num_base_priors = [1] * 5
        core = _stub("mmdet.core", multi_apply=multi_apply, build_prior_generator=lambda cfg: _Prior(),
                     build_anchor_generator=lambda cfg: _Prior(), build_bbox_coder=lambda cfg: object())
        for nm in ("anchor_inside_flags", "bbox_overlaps", "build_assigner", "build_sampler", "images_to_levels", "reduce_mean", "unmap",
                   "multiclass_nms", "bbox_mapping_back", "merge_aug_proposals"):
            setattr(core, nm, None)
        _stub("mmdet.core.utils", filter_scores_and_topk=None, select_single_mlvl=None)
        _stub("mmdet.utils")
        _stub("mmdet.utils.contextmanagers", completed=None)      # (async test path of dense_test_mixins.py, python >= 3.7 only)
        for pkg in ("mmdet", "mmdet.models", "mmdet.models.dense_heads"):
            m = _stub(pkg)
            m.__path__ = []
        reg = type("Registry", (), {"register_module": lambda self, *a, **k: (lambda c: c)})()
        _stub("mmdet.models.builder", HEADS=reg, build_loss=lambda cfg: nn.Identity())
        root = os.path.join("/root/reference", "yolox-ufp", "mmdet", "models", "dense_heads")
        for f in ("base_dense_head", "dense_test_mixins", "anchor_head", "gfl_head", "mp_head"):
            _load_ref_module("mmdet.models.dense_heads." + f, os.path.join(root, f + ".py"))
        gfl = sys.modules["mmdet.models.dense_heads.gfl_head"]
        mp = sys.modules["mmdet.models.dense_heads.mp_head"]
        out = {}
        block = make_block(out)
        common = dict(num_classes=10, in_channels=256, feat_channels=256, stacked_convs=4,
                      anchor_generator=dict(type="AnchorGenerator", ratios=[1.0], octave_base_scale=8, scales_per_octave=1,
                                            strides=[8, 16, 32, 64, 128]),
                      loss_cls=dict(type="QualityFocalLoss", use_sigmoid=True, beta=2.0, loss_weight=1.0),
                      loss_dfl=dict(type="DistributionFocalLoss", loss_weight=0.25), reg_max=16,
                      loss_bbox=dict(type="GIoULoss", loss_weight=2.0))

        def levels_of(x):                                  # five pyramid levels from one seeded tensor (mirrored by the tests)
            f = [x]
            for _ in range(4):
                f.append(torch.nn.functional.avg_pool2d(f[-1], 2, ceil_mode=True))
            return f

        def wrap(ctor):
            class Head(nn.Module):
                def __init__(self):
                    super().__init__()
                    self.bbox_head = ctor()

                def forward(self, x):
                    cls, reg = self.bbox_head(levels_of(x))
                    return torch.cat([t.flatten(1) for t in cls] + [t.flatten(1) for t in reg], 1)
            return Head
        block("gfl_head_forward", wrap(lambda: gfl.GFLHead(**common)), (2, 256, 24, 40))
        block("mp_head_forward", wrap(lambda: mp.MPHead(num_words=8, gamma=10, proxies_list=[2, 3, 2, 5, 4, 8, 8, 4, 3, 3], **common)),
              (2, 256, 20, 28), seed=1)
        x = synth_input((37, 68), 5) * 3.0
        out["integral/x"] = x.numpy()
        out["integral/y"] = gfl.Integral(16)(x).numpy()

        # ---- get_bboxes up to (not including) the NMS: base_dense_head.get_bboxes -> gfl_head._get_bboxes_single(with_nms=False)
        # with the reference's REAL AnchorGenerator, DistancePointBBoxCoder / distance2bbox, filter_scores_and_topk and
        # select_single_mlvl (loaded by path; only their registries and mask imports are stood in).  mmcv.ops.batched_nms is a
        # compiled op: the NMS itself stays unpinned, exactly as torchvision's for the YOLOX path.
        core_root = os.path.join("/root/reference", "yolox-ufp", "mmdet", "core")
        for pkg in ("mmdet.core.anchor", "mmdet.core.bbox", "mmdet.core.bbox.coder", "mmdet.core.mask"):
            m = _stub(pkg)
            m.__path__ = []
        _stub("mmdet.core.anchor.builder", PRIOR_GENERATORS=reg, ANCHOR_GENERATORS=reg)
        _stub("mmdet.core.bbox.builder", BBOX_CODERS=reg)
        _stub("mmdet.core.mask.structures", BitmapMasks=type("BitmapMasks", (), {}), PolygonMasks=type("PolygonMasks", (), {}))
        sys.modules["mmcv"].is_tuple_of = lambda seq, t: isinstance(seq, tuple) and all(isinstance(v, t) for v in seq)
        ag = _load_ref_module("mmdet.core.anchor.anchor_generator", os.path.join(core_root, "anchor", "anchor_generator.py"))
        _load_ref_module("mmdet.core.bbox.transforms", os.path.join(core_root, "bbox", "transforms.py"))
        _load_ref_module("mmdet.core.bbox.coder.base_bbox_coder", os.path.join(core_root, "bbox", "coder", "base_bbox_coder.py"))
        dc = _load_ref_module("mmdet.core.bbox.coder.distance_point_bbox_coder",
                              os.path.join(core_root, "bbox", "coder", "distance_point_bbox_coder.py"))
        misc = _load_ref_module("mmdet.core.utils.misc", os.path.join(core_root, "utils", "misc.py"))
        tr = sys.modules["mmdet.core.bbox.transforms"]              # bbox2result (transforms.py:116-133): the result format
        rng2 = np.random.default_rng(3)
        bb = np.concatenate([rng2.uniform(0, 100, (23, 4)), rng2.uniform(0, 1, (23, 1))], 1).astype(np.float32)
        lb = rng2.integers(0, 10, 23).astype(np.int64)
        out["bbox2result/boxes"], out["bbox2result/labels"] = bb, lb
        for c, arr in enumerate(tr.bbox2result(torch.from_numpy(bb), torch.from_numpy(lb), 10)):
            out["bbox2result/class%d" % c] = arr
        for c, arr in enumerate(tr.bbox2result(np.zeros((0, 5), np.float32), np.zeros((0,), np.int64), 3)):
            out["bbox2result/empty%d" % c] = arr
        bdh = sys.modules["mmdet.models.dense_heads.base_dense_head"]
        gfl.filter_scores_and_topk = misc.filter_scores_and_topk
        bdh.filter_scores_and_topk, bdh.select_single_mlvl = misc.filter_scores_and_topk, misc.select_single_mlvl

        class Cfg(dict):
            __getattr__ = dict.get
        head = gfl.GFLHead(**common)
        head.prior_generator = ag.AnchorGenerator(strides=[8, 16, 32, 64, 128], ratios=[1.0], octave_base_scale=8, scales_per_octave=1)
        head.bbox_coder = dc.DistancePointBBoxCoder()
        head.eval()
        sizes = [(12, 20), (6, 10), (3, 5), (2, 3), (1, 2)]
        rng = np.random.default_rng(11)
        cls = [torch.from_numpy(rng.normal(-2.0, 1.5, (2, 10) + s).astype(np.float32)) for s in sizes]
        reg = [torch.from_numpy(rng.normal(0.0, 2.0, (2, 68) + s).astype(np.float32)) for s in sizes]
        metas = [dict(img_shape=(90, 150, 3), scale_factor=np.array([1.25, 1.25, 1.25, 1.25], np.float32)),
                 dict(img_shape=(96, 160, 3), scale_factor=np.array([0.8, 0.75, 0.8, 0.75], np.float32))]
        for i, l in enumerate(cls):
            out["bboxes/cls/%d" % i] = l.numpy()
            out["bboxes/reg/%d" % i] = reg[i].numpy()
        for tag, rescale, cfg in (("plain", False, Cfg(nms_pre=1000, score_thr=0.05, max_per_img=100)),
                                  ("topk_rescale", True, Cfg(nms_pre=40, score_thr=0.2, max_per_img=100))):
            res = head.get_bboxes(cls, reg, img_metas=metas, cfg=cfg, rescale=rescale, with_nms=False)
            for b, (bx, sc, lb) in enumerate(res):
                out["bboxes/%s/%d/boxes" % (tag, b)] = bx.numpy()
                out["bboxes/%s/%d/scores" % (tag, b)] = sc.numpy()
                out["bboxes/%s/%d/labels" % (tag, b)] = lb.numpy()
        np.savez_compressed(os.path.join(HERE, "head_golden.npz"), **out)
        print("heads:", len(out), "arrays,", os.path.getsize(os.path.join(HERE, "head_golden.npz")), "bytes")
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def yolox_mmdet_cases():
    """The mmdet flavour of the YOLOX path (SURVEY 8a note: yolox-ufp/mmdet CSPDarknet / YOLOXPAFPN / YOLOXHead are the same
    network as yolox-drone's under other parameter names): the reference's OWN backbones/csp_darknet.py, utils/csp_layer.py,
    necks/yolox_pafpn.py, dense_heads/yolox_head.py (+ base_dense_head, dense_test_mixins, core/anchor/point_generator.py),
    loaded by path.  Stand-ins: ConvModule = conv (no bias under a norm) -> BatchNorm2d(eps, momentum of the cfg) -> Swish
    under mmcv's attribute names `conv`, `bn`, `activate`; BaseModule; registries; constructor-only dummies for the losses.
    Pins the module structure and parameter NAMES the mmdet_surface twins and `mmdet_to_drone_key` must reproduce, the raw
    head outputs, and `_bbox_decode` on the reference's own priors."""
    names = ["mmcv", "mmcv.cnn", "mmcv.runner", "mmcv.ops", "mmcv.ops.nms", "mmcv.cnn.utils", "mmcv.cnn.utils.weight_init", "mmdet",
             "mmdet.core", "mmdet.core.utils", "mmdet.core.anchor", "mmdet.core.anchor.builder", "mmdet.core.anchor.point_generator",
             "mmdet.utils", "mmdet.utils.contextmanagers", "mmdet.models", "mmdet.models.builder", "mmdet.models.utils",
             "mmdet.models.utils.csp_layer", "mmdet.models.backbones", "mmdet.models.backbones.csp_darknet", "mmdet.models.necks",
             "mmdet.models.necks.yolox_pafpn", "mmdet.models.dense_heads", "mmdet.models.dense_heads.base_dense_head",
             "mmdet.models.dense_heads.dense_test_mixins", "mmdet.models.dense_heads.yolox_head"]
    saved = {k: sys.modules.get(k) for k in names}
    try:
        mmcv = _mmcv_building_blocks()
        nn = torch.nn

        class Swish(nn.Module):
            def forward(self, x):
                return x * torch.sigmoid(x)

        class ConvModule(nn.Module):
            def __init__(self, cin, cout, k, stride=1, padding=0, dilation=1, groups=1, bias="auto", conv_cfg=None, norm_cfg=None,
                         act_cfg=dict(type="ReLU"), inplace=True):
                super().__init__()
                assert conv_cfg is None and norm_cfg["type"] == "BN" and act_cfg["type"] == "Swish" and bias == "auto"
                self.conv = nn.Conv2d(cin, cout, k, stride, padding, dilation, groups, bias=False)
                self.bn = nn.BatchNorm2d(cout, eps=norm_cfg.get("eps", 1e-5), momentum=norm_cfg.get("momentum", 0.1))
                self.activate = Swish()

            def forward(self, x):
                return self.activate(self.bn(self.conv(x)))
        class DepthwiseSeparableConvModule(nn.Module):
            """mmcv.cnn.DepthwiseSeparableConvModule restated (mmcv is not installable here): a depthwise ConvModule
            (groups = in_channels, the caller's kernel / stride / padding) followed by a pointwise 1x1 ConvModule, both with
            the caller's norm_cfg / act_cfg ('default' for the dw_* / pw_* overrides), attribute names as upstream."""
            def __init__(self, cin, cout, k, stride=1, padding=0, dilation=1, norm_cfg=None, act_cfg=dict(type="ReLU"), **kw):
                super().__init__()
                self.depthwise_conv = ConvModule(cin, cin, k, stride=stride, padding=padding, dilation=dilation, groups=cin,
                                                 norm_cfg=norm_cfg, act_cfg=act_cfg, **kw)
                self.pointwise_conv = ConvModule(cin, cout, 1, norm_cfg=norm_cfg, act_cfg=act_cfg, **kw)

            def forward(self, x):
                return self.pointwise_conv(self.depthwise_conv(x))
        noop = lambda *a, **k: None
        mmcv.cnn.ConvModule, mmcv.cnn.DepthwiseSeparableConvModule, mmcv.cnn.bias_init_with_prob = ConvModule, DepthwiseSeparableConvModule, (lambda p_: 0.0)
        _stub("mmcv.cnn.utils")
        _stub("mmcv.cnn.utils.weight_init", constant_init=noop)
        _stub("mmcv.ops", batched_nms=None)
        _stub("mmcv.ops.nms", batched_nms=None)
        for pkg in ("mmdet", "mmdet.models", "mmdet.models.utils", "mmdet.models.backbones", "mmdet.models.necks",
                    "mmdet.models.dense_heads", "mmdet.core.anchor"):
            m = _stub(pkg)
            m.__path__ = []
        reg = type("Registry", (), {"register_module": lambda self, *a, **k: (lambda c: c)})()
        _stub("mmdet.models.builder", BACKBONES=reg, NECKS=reg, HEADS=reg, build_loss=lambda cfg: nn.Identity())
        _stub("mmdet.core.anchor.builder", PRIOR_GENERATORS=reg)
        _stub("mmdet.utils")
        _stub("mmdet.utils.contextmanagers", completed=None)
        core_root = os.path.join("/root/reference", "yolox-ufp", "mmdet", "core")
        pg = _load_ref_module("mmdet.core.anchor.point_generator", os.path.join(core_root, "anchor", "point_generator.py"))

        def multi_apply(func, *args, **kwargs):
            from functools import partial
            pfunc = partial(func, **kwargs) if kwargs else func
            return tuple(map(list, zip(*map(pfunc, *args))))
        core = _stub("mmdet.core", MlvlPointGenerator=pg.MlvlPointGenerator, multi_apply=multi_apply)
        for nm in ("bbox_xyxy_to_cxcywh", "build_assigner", "build_sampler", "reduce_mean", "bbox_mapping_back", "merge_aug_proposals"):
            setattr(core, nm, None)
        _stub("mmdet.core.utils", filter_scores_and_topk=None, select_single_mlvl=None)
        root = os.path.join("/root/reference", "yolox-ufp", "mmdet", "models")
        csp = _load_ref_module("mmdet.models.utils.csp_layer", os.path.join(root, "utils", "csp_layer.py"))
        sys.modules["mmdet.models.utils"].CSPLayer = csp.CSPLayer
        bb = _load_ref_module("mmdet.models.backbones.csp_darknet", os.path.join(root, "backbones", "csp_darknet.py"))
        nk = _load_ref_module("mmdet.models.necks.yolox_pafpn", os.path.join(root, "necks", "yolox_pafpn.py"))
        for f in ("base_dense_head", "dense_test_mixins", "yolox_head"):
            _load_ref_module("mmdet.models.dense_heads." + f, os.path.join(root, "dense_heads", f + ".py"))
        yh = sys.modules["mmdet.models.dense_heads.yolox_head"]
        out = {}
        block = make_block(out)

        class Det(nn.Module):        # configs/yolox/yolox_s_8x8_300e_coco.py:6-22 with 10 classes
            def __init__(self):
                super().__init__()
                self.backbone = bb.CSPDarknet(deepen_factor=0.33, widen_factor=0.5)
                self.neck = nk.YOLOXPAFPN(in_channels=[128, 256, 512], out_channels=128, num_csp_blocks=1)
                self.bbox_head = yh.YOLOXHead(num_classes=10, in_channels=128, feat_channels=128)

            def forward(self, x):
                cls, reg, obj = self.bbox_head(self.neck(self.backbone(x)))
                return torch.cat([torch.cat((r, o, c), 1).flatten(1) for c, r, o in zip(cls, reg, obj)], 1)
        class DetNano(nn.Module):    # configs/yolox/yolox_nano_8x8_300e_coco.py:4-11 (use_depthwise=True everywhere) with 10 classes
            def __init__(self):
                super().__init__()
                self.backbone = bb.CSPDarknet(deepen_factor=0.33, widen_factor=0.25, use_depthwise=True)
                self.neck = nk.YOLOXPAFPN(in_channels=[64, 128, 256], out_channels=64, num_csp_blocks=1, use_depthwise=True)
                self.bbox_head = yh.YOLOXHead(num_classes=10, in_channels=64, feat_channels=64, use_depthwise=True)

            def forward(self, x):
                cls, reg, obj = self.bbox_head(self.neck(self.backbone(x)))
                return torch.cat([torch.cat((r, o, c), 1).flatten(1) for c, r, o in zip(cls, reg, obj)], 1)
        with torch.no_grad():
            block("yolox_s_mmdet", Det, (2, 3, 96, 128), calibrate=True)
            block("yolox_nano_mmdet", DetNano, (2, 3, 96, 128), seed=1, calibrate=True)
            head = yh.YOLOXHead(num_classes=10, in_channels=128, feat_channels=128)
            priors = torch.cat(head.prior_generator.grid_priors([(12, 16), (6, 8), (3, 4)], device="cpu", with_stride=True))
            preds = synth_input((2, priors.shape[0], 4), 9)
            out["decode/preds"] = preds.numpy()
            out["decode/priors"] = priors.numpy()
            out["decode/boxes"] = head._bbox_decode(priors, preds).numpy()
        np.savez_compressed(os.path.join(HERE, "yolox_mmdet_golden.npz"), **out)
        print("yolox mmdet:", len(out), "arrays,", os.path.getsize(os.path.join(HERE, "yolox_mmdet_golden.npz")), "bytes")
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def main():
    if "--merge-eval-only" in sys.argv:
        merge_cases()
        eval_cases()
        return
    if "--resdet-only" in sys.argv:
        resdet_cases()
        head_cases()
        yolox_mmdet_cases()
        return
    if "--yolox-mmdet-only" in sys.argv:
        yolox_mmdet_cases()
        return
    if "--attention-only" in sys.argv:
        att = {}
        attention_cases(make_block(att))
        np.savez_compressed(os.path.join(HERE, "attention_golden.npz"), **att)
        print("attention:", len(att), "arrays,", os.path.getsize(os.path.join(HERE, "attention_golden.npz")), "bytes")
        return
    merge_cases()
    eval_cases()
    ufp_cases()
    preprocess_cases()
    from models.base import yolox as ref_base
    from models.base.baseConv import BaseConv, DWConv
    from models.base.darknet import Bottleneck, CSPLayer, Focus, SPPBottleneck
    from models.block.non_local import yolo_patch_nonlocal_plus as ref_gl
    from models.block.non_local.Identity_Conv import (Identity_Conv_five, Identity_Conv_seven,
                                                      Identity_Conv_three, Non_local_Block,
                                                      Patch_Conv, Patch_Conv_NonLocal)
    # decode_outputs is pure torch but its module imports torchvision (absent) at the top:
    # register an empty stand-in so the import succeeds; NMS itself is never called here.
    if "torchvision" not in sys.modules:
        tv = types.ModuleType("torchvision")
        ops = types.ModuleType("torchvision.ops")
        ops.boxes = types.ModuleType("torchvision.ops.boxes")
        tv.ops = ops
        sys.modules.update({"torchvision": tv, "torchvision.ops": ops,
                            "torchvision.ops.boxes": ops.boxes})
    from models.core.utils_bbox import decode_outputs, yolo_correct_boxes

    out = {}

    # ------------------------------------------------------------------ blocks
    block = make_block(out)

    for k in (1, 3):
        for s in (1, 2):
            for act in ("silu", "relu", "lrelu"):
                block("baseconv_k%d_s%d_%s" % (k, s, act),
                      lambda k=k, s=s, act=act: BaseConv(16, 32, k, s, act=act), (2, 16, 20, 24))
    block("dwconv_k3_s2", lambda: DWConv(16, 32, 3, 2), (2, 16, 20, 24))
    block("focus", lambda: Focus(3, 16, 3), (2, 3, 32, 40))
    block("spp", lambda: SPPBottleneck(32, 32), (2, 32, 20, 24))
    block("bottleneck_add", lambda: Bottleneck(16, 16, True, 1.0), (2, 16, 20, 24))
    block("bottleneck_noadd", lambda: Bottleneck(16, 16, False, 1.0), (2, 16, 20, 24))
    block("csp_n2_shortcut", lambda: CSPLayer(32, 32, 2, True), (2, 32, 20, 24))
    block("csp_n1_noshortcut", lambda: CSPLayer(48, 32, 1, False), (2, 48, 20, 24))
    block("nonlocal_c16", lambda: Non_local_Block(16, 16), (2, 16, 10, 12))
    block("nonlocal_c32_inter16", lambda: Non_local_Block(32, None), (2, 32, 5, 7))
    block("patch_conv_s1", lambda: Patch_Conv(32, 16, patch_scale=4, stride=1), (2, 32, 20, 24))
    block("patch_conv_nonlocal_s2", lambda: Patch_Conv_NonLocal(16, 32, patch_scale=2), (2, 16, 40, 48))
    block("identity3", lambda: Identity_Conv_three(16, 16), (2, 16, 20, 24))
    block("identity5", lambda: Identity_Conv_five(16, 16), (2, 16, 20, 24))
    block("identity7", lambda: Identity_Conv_seven(16, 16), (2, 16, 20, 24))

    # ------------------------------------------------------------------ cross-scale decoupled head
    # drone/models/lsk/yolox6.py:7-153 (text-identical to new/yolox6.py): the head class alone,
    # fed with four synthetic pyramid features (dark2, P3, P4, P5 of a width-0.5 model).
    from models.lsk.yolox6 import YOLOXHead as CrossHead
    for seed in (0, 1):
        m = CrossHead(10, 0.5)
        shapes = fill(m, seed)
        feats = [synth_input(sh, seed + 300 + i) for i, sh in enumerate(
            [(2, 64, 32, 40), (2, 128, 16, 20), (2, 256, 8, 10), (2, 512, 4, 5)])]
        ys = m(feats)
        for i, y in enumerate(ys):
            out["crosshead/seed%d/out%d" % (seed, i)] = y.numpy()
        out["crosshead/seed%d/meta" % seed] = np.frombuffer(json.dumps(
            {"shapes": shapes, "seed": seed, "feat_shapes": [list(f.shape) for f in feats]}).encode(), np.uint8)

    # ------------------------------------------------------------------ whole models
    shapes_all = {}
    for mname, mod in (("base", ref_base), ("gl", ref_gl)):
        for phi in ("nano", "tiny", "s"):
            for seed in (0, 1):
                if phi == "s" and seed == 1:
                    continue
                net = mod.YoloBody(10, phi)
                shapes = fill(net, seed)
                shapes_all["%s_%s" % (mname, phi)] = shapes
                in_shape = (2, 3, 128, 160) if phi != "s" else (1, 3, 128, 160)
                x = synth_input(in_shape, seed + 100)
                tag = "model/%s_%s_seed%d" % (mname, phi, seed)
                for k, v in calibrate_bn(net, x, seed).items():
                    out["%s/bn/%s" % (tag, k)] = v
                ys = net(x)
                for i, y in enumerate(ys):
                    out["%s/out%d" % (tag, i)] = y.numpy()
                dec = decode_outputs([y.clone() for y in ys], list(in_shape[2:]))
                out["%s/decoded" % tag] = dec.numpy()
                out["%s/meta" % tag] = np.frombuffer(json.dumps(
                    {"in_shape": list(in_shape), "seed": seed, "num_classes": 10, "phi": phi,
                     "model": mname}).encode(), np.uint8)
                print(tag, [tuple(y.shape) for y in ys],
                      "logit std %.3f" % float(torch.cat([y.flatten() for y in ys]).std()))

    # ------------------------------------------------------------------ yolo_correct_boxes
    rng = np.random.default_rng(7)
    xy = rng.uniform(0.1, 0.9, (32, 2)).astype(np.float32)
    wh = rng.uniform(0.01, 0.3, (32, 2)).astype(np.float32)
    for lb in (False, True):
        out["correct_boxes/letterbox%d" % int(lb)] = yolo_correct_boxes(
            xy.copy(), wh.copy(), [640, 640], np.array([540, 1024]), lb)
    out["correct_boxes/xy"], out["correct_boxes/wh"] = xy, wh

    attention_cases(block)
    att = {k: out.pop(k) for k in list(out) if k.startswith("block/att_") or k.startswith("attnet/")}
    np.savez_compressed(os.path.join(HERE, "attention_golden.npz"), **att)
    np.savez_compressed(os.path.join(HERE, "drone_golden.npz"), **out)
    with open(os.path.join(HERE, "shapes.json"), "w") as f:
        json.dump(shapes_all, f, separators=(",", ":"))
    print("wrote", len(out), "arrays;", os.path.getsize(os.path.join(HERE, "drone_golden.npz")), "bytes")
    resdet_cases()              # yolox-ufp/mmdet: ResNet + FPN, GFLHead / MPHead / get_bboxes, the mmdet flavour of YOLOX
    head_cases()
    yolox_mmdet_cases()


if __name__ == "__main__":
    main()
