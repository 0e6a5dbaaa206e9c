"""Architecture tables: the parameter/buffer names and shapes of the reference's detectors,
generated from (kind, phi, num_classes) so that our modules expose EXACTLY the reference's
state_dict (a reference checkpoint loads with `load_state_dict`, and vice versa).

Sources: drone/models/base/{baseConv,darknet,yolox}.py,
drone/models/block/non_local/{Identity_Conv,yolo_patch_nonlocal_plus}.py (key names are
the attribute paths of those modules; order = their registration order).
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Tuple

# drone/models/base/yolox.py:240-241
DEPTH = {"nano": 0.33, "tiny": 0.33, "s": 0.33, "m": 0.67, "l": 1.00, "x": 1.33}
WIDTH = {"nano": 0.25, "tiny": 0.375, "s": 0.50, "m": 0.75, "l": 1.00, "x": 1.25}
KINDS = ("base", "gl", "cross")


class _Table(OrderedDict):
    def conv_bn(self, p: str, cin: int, cout: int, k: int, groups: int = 1):
        self[p + ".conv.weight"] = (cout, cin // groups, k, k)
        self[p + ".bn.weight"] = (cout,)
        self[p + ".bn.bias"] = (cout,)
        self[p + ".bn.running_mean"] = (cout,)
        self[p + ".bn.running_var"] = (cout,)
        self[p + ".bn.num_batches_tracked"] = ()

    def any_conv(self, p: str, cin: int, cout: int, k: int, depthwise: bool):
        if depthwise:                       # DWConv: depthwise kxk then pointwise 1x1
            self.conv_bn(p + ".dconv", cin, cin, k, groups=cin)
            self.conv_bn(p + ".pconv", cin, cout, 1)
        else:
            self.conv_bn(p, cin, cout, k)

    def plain(self, p: str, cin: int, cout: int, k: int):
        self[p + ".weight"] = (cout, cin, k, k)
        self[p + ".bias"] = (cout,)

    def csp(self, p: str, cin: int, cout: int, n: int, depthwise: bool):
        hid = int(cout * 0.5)
        self.conv_bn(p + ".conv1", cin, hid, 1)
        self.conv_bn(p + ".conv2", cin, hid, 1)
        self.conv_bn(p + ".conv3", 2 * hid, cout, 1)
        for i in range(n):
            self.conv_bn("%s.m.%d.conv1" % (p, i), hid, hid, 1)
            self.any_conv("%s.m.%d.conv2" % (p, i), hid, hid, 3, depthwise)

    def attention(self, p: str, d: int):
        """Attention (drone/models/new/Non_local_family.py:254-263), gating unit channel_scale=1."""
        self.plain(p + ".proj_1", d, d, 1)
        g = p + ".spatial_gating_unit"
        for q in ("lt", "lb", "rt", "rb"):
            self.nonlocal_block("%s.feat_patchconv_%s_nonlocal" % (g, q), d, d)
        self.conv_bn(g + ".channel_conv", d, d, 3)
        self.plain(p + ".proj_2", d, d, 1)

    def lsk_attention(self, p: str, d: int):
        """Attention with an LSKblock (drone/models/lsk/LSK.py:27-71), registration order of the two __init__s."""
        self.plain(p + ".proj_1", d, d, 1)
        g = p + ".spatial_gating_unit"
        for q, k in ((".conv0", 5), (".conv_spatial", 7)):          # depthwise: groups = dim
            self[g + q + ".weight"] = (d, 1, k, k)
            self[g + q + ".bias"] = (d,)
        self.plain(g + ".conv1", d, d // 2, 1)
        self.plain(g + ".conv2", d, d // 2, 1)
        self.plain(g + ".conv_squeeze", 2, 2, 7)
        self.plain(g + ".conv", d // 2, d, 1)
        self.plain(p + ".proj_2", d, d, 1)

    def darknet(self, p: str, dep: float, wid: float, depthwise: bool, attention=False):
        base, depth = int(wid * 64), max(round(dep * 3), 1)
        self.conv_bn(p + ".stem.conv", 12, base, 3)
        for i, (name, mult, n) in enumerate((("dark2", 2, depth), ("dark3", 4, depth * 3), ("dark4", 8, depth * 3))):
            cin = base * mult // 2
            self.any_conv("%s.%s.0" % (p, name), cin, base * mult, 3, depthwise)
            self.csp("%s.%s.1" % (p, name), base * mult, base * mult, n, depthwise)
        self.any_conv(p + ".dark5.0", base * 8, base * 16, 3, depthwise)
        self.conv_bn(p + ".dark5.1.conv1", base * 16, base * 8, 1)
        self.conv_bn(p + ".dark5.1.conv2", base * 8 * 4, base * 16, 1)
        self.csp(p + ".dark5.2", base * 16, base * 16, depth, depthwise)
        if attention:                       # drone/models/new/darknet_att.py:161-164; "lsk": drone/models/lsk/darknet_lsk.py
            for i, mult in enumerate((2, 4, 8, 16)):
                (self.lsk_attention if attention == "lsk" else self.attention)("%s.lsk%d" % (p, i + 2), base * mult)

    def nonlocal_block(self, p: str, cin: int, ci: int):
        self.plain(p + ".g", cin, ci, 1)
        self.plain(p + ".theta", cin, ci, 1)
        self.plain(p + ".phi", cin, ci, 1)
        self.plain(p + ".conv_out", ci, cin, 1)

    def patch_conv(self, p: str, cin: int, cout: int, with_nonlocal: bool):
        mid = int(0.5 * cin)
        for q in ("lt", "lb", "rt", "rb"):
            self.conv_bn("%s.feat_patchconv_%s" % (p, q), cin, mid, 3)
        if with_nonlocal:
            for q in ("lt", "lb", "rt", "rb"):
                self.nonlocal_block("%s.feat_patchconv_%s_nonlocal" % (p, q), mid, mid)
        for q in ("r", "l", "t", "b"):
            self.conv_bn("%s.feat_patchconv_%s" % (p, q), mid, mid, 3)
        self.plain(p + ".channel_conv", 2 * mid, cout, 1)


def cross_head_table(t: "_Table", h: str, num_classes: int, wid: float, dw: bool):
    """Cross-scale decoupled head, drone/models/lsk/yolox6.py:7-67 (= new/yolox6.py)."""
    c = [int(256 * wid), int(512 * wid), int(1024 * wid)]
    f = int(256 * wid)
    for i in range(3):
        cin = f * 2 if i == 2 else f * 3
        t.any_conv("%s.cls_convs.%d.0" % (h, i), cin, cin, 3, dw)
        t.any_conv("%s.cls_convs.%d.1" % (h, i), cin, f, 3, dw)
    for i in range(3):
        for j in range(2):
            t.any_conv("%s.reg_convs.%d.%d" % (h, i, j), f, f, 3, dw)
    for i in range(3):
        t.plain("%s.cls_preds.%d" % (h, i), f, num_classes, 1)
    for i in range(3):
        t.plain("%s.reg_preds.%d" % (h, i), f, 4, 1)
    for i in range(3):
        t.plain("%s.obj_preds.%d" % (h, i), f, 1, 1)
    for i in range(3):
        t.conv_bn("%s.stems.%d" % (h, i), c[i], f, 1)
    t.csp(h + ".csp_feat0", int(0.5 * 256 * wid), f, round(3 * 0.75), dw)
    for i in range(3):
        t.any_conv("%s.up_convs.%d.0" % (h, i), f, f, 3, dw)
        t.any_conv("%s.up_convs.%d.1" % (h, i), f, f, 3, dw)


def state_dict_shapes(kind: str, phi: str, num_classes: int,
                      attention_backbone=False) -> "OrderedDict[str, Tuple[int, ...]]":
    """attention_backbone=True: the backbone is new/darknet_att.py's CSPDarknet (an Attention
    block after each stage) instead of base/darknet.py's; "lsk": lsk/darknet_lsk.py's (the same with LSK.Attention)."""
    if kind not in KINDS:
        raise ValueError("kind must be one of %r" % (KINDS,))
    if phi not in DEPTH:
        raise KeyError(phi)
    dep, wid = DEPTH[phi], WIDTH[phi]
    dw = phi == "nano"
    c = [int(256 * wid), int(512 * wid), int(1024 * wid)]
    n = round(3 * dep)
    t = _Table()
    b = "backbone"
    t.darknet(b + ".backbone", dep, wid, dw, attention_backbone)
    t.conv_bn(b + ".lateral_conv0", c[2], c[1], 1)
    t.csp(b + ".C3_p4", (3 if kind == "gl" else 2) * c[1], c[1], n, dw)
    t.conv_bn(b + ".reduce_conv1", c[1], c[0], 1)
    t.csp(b + ".C3_p3", 2 * c[0], c[0], n, dw)
    if kind == "gl":
        t.plain(b + ".P3_Identity.conv", c[0], c[0], 7)
    t.any_conv(b + ".bu_conv2", c[0], c[0], 3, dw)
    t.csp(b + ".C3_n3", (3 if kind == "gl" else 2) * c[0], c[1], n, dw)
    if kind == "gl":
        t.plain(b + ".P4_Identity.conv", c[1], c[1], 5)
    t.any_conv(b + ".bu_conv1", c[1], c[1], 3, dw)
    t.csp(b + ".C3_n4", 2 * c[1], c[2], n, dw)
    if kind == "gl":
        t.patch_conv(b + ".Patch_conv_feat1", c[0], c[1], True)
        t.patch_conv(b + ".Patch_conv_feat2", c[1], c[0], False)
        t.plain(b + ".P5_Identity.conv", c[2], c[2], 3)
    f = int(256 * wid)
    h = "head"
    if kind == "cross":
        cross_head_table(t, h, num_classes, wid, dw)
        return t
    for name in ("cls_convs", "reg_convs"):
        for i in range(3):
            for j in range(2):
                t.any_conv("%s.%s.%d.%d" % (h, name, i, j), f, f, 3, dw)
    for i in range(3):
        t.plain("%s.cls_preds.%d" % (h, i), f, num_classes, 1)
    for i in range(3):
        t.plain("%s.reg_preds.%d" % (h, i), f, 4, 1)
    for i in range(3):
        t.plain("%s.obj_preds.%d" % (h, i), f, 1, 1)
    for i in range(3):
        t.conv_bn("%s.stems.%d" % (h, i), c[i], f, 1)
    return t


# ------------------------------------------------------------------------------------------
# ResNet + FPN + GFLHead / MPHead (mmdet key names)
RESNET_STAGE_BLOCKS = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3)}       # ufp/mmdet/models/backbones/resnet.py:363-369


def _bn(t: "_Table", p: str, c: int):
    t[p + ".weight"] = (c,)
    t[p + ".bias"] = (c,)
    t[p + ".running_mean"] = (c,)
    t[p + ".running_var"] = (c,)
    t[p + ".num_batches_tracked"] = ()


def resnet_table(t: "_Table", p: str, depth: int = 50, base: int = 64):
    """torchvision-style names, resnet.py:571-598 (stem) and res_layer.py (stages)."""
    t[p + ".conv1.weight"] = (base, 3, 7, 7)
    _bn(t, p + ".bn1", base)
    inplanes = base
    for i, nblocks in enumerate(RESNET_STAGE_BLOCKS[depth]):
        planes = base * 2 ** i
        for j in range(nblocks):
            q = "%s.layer%d.%d" % (p, i + 1, j)
            t[q + ".conv1.weight"] = (planes, inplanes, 1, 1)
            _bn(t, q + ".bn1", planes)
            t[q + ".conv2.weight"] = (planes, planes, 3, 3)
            _bn(t, q + ".bn2", planes)
            t[q + ".conv3.weight"] = (planes * 4, planes, 1, 1)
            _bn(t, q + ".bn3", planes * 4)
            if j == 0:                      # stride != 1 or inplanes != planes*4 (res_layer.py:40)
                t[q + ".downsample.0.weight"] = (planes * 4, inplanes, 1, 1)
                _bn(t, q + ".downsample.1", planes * 4)
            inplanes = planes * 4


def fpn_table(t: "_Table", p: str, in_channels, out_channels: int, start_level: int, num_outs: int, add_extra_convs):
    """fpn.py:105-148: all lateral convs, then all fpn convs (incl. the extra stride-2 ones)."""
    n_lat = len(in_channels) - start_level
    for i in range(n_lat):
        t.plain("%s.lateral_convs.%d.conv" % (p, i), in_channels[i + start_level], out_channels, 1)
    for i in range(n_lat):
        t.plain("%s.fpn_convs.%d.conv" % (p, i), out_channels, out_channels, 3)
    if add_extra_convs:
        mode = "on_input" if add_extra_convs is True else add_extra_convs
        for i in range(num_outs - n_lat):
            cin = in_channels[-1] if (i == 0 and mode == "on_input") else out_channels
            t.plain("%s.fpn_convs.%d.conv" % (p, n_lat + i), cin, out_channels, 3)


def gl_fusion_table(t: "_Table", p: str, channels: int, channel_cat: str = "linear"):
    """Patch_Conv_NonLocal_new(in_channel=C, out_channel=C, channel_scale=1, patch_scale=2, channel_cat)
    (drone/models/new/Non_local_family.py:208-228), registration order of its __init__: four Non_local_Blocks
    (g, theta, phi, conv_out: :15-18) with inter_channels = int(channel_scale * C) = C, then channel_conv."""
    for q in ("lt", "lb", "rt", "rb"):
        for nm in ("g", "theta", "phi"):
            t.plain("%s.feat_patchconv_%s_nonlocal.%s" % (p, q, nm), channels, channels, 1)
        t.plain("%s.feat_patchconv_%s_nonlocal.conv_out" % (p, q), channels, channels, 1)
    if channel_cat == "linear":
        t.plain(p + ".channel_conv", channels, channels, 1)
    else:                                   # BaseConv(C, C, 3, 1, act='silu')
        t[p + ".channel_conv.conv.weight"] = (channels, channels, 3, 3)
        _bn(t, p + ".channel_conv.bn", channels)


def _gn_tower(t: "_Table", p: str, cin: int, feat: int, stacked: int):
    for name in ("cls_convs", "reg_convs"):
        for i in range(stacked):
            q = "%s.%s.%d" % (p, name, i)
            t[q + ".conv.weight"] = (feat, cin if i == 0 else feat, 3, 3)
            t[q + ".gn.weight"] = (feat,)
            t[q + ".gn.bias"] = (feat,)


def gfl_head_table(t: "_Table", p: str, num_classes: int, in_channels: int = 256, feat: int = 256, stacked: int = 4,
                   reg_max: int = 16, n_levels: int = 5):
    """gfl_head.py:128-152 (+ Integral buffer :32-33)."""
    _gn_tower(t, p, in_channels, feat, stacked)
    t.plain(p + ".gfl_cls", feat, num_classes, 3)
    t.plain(p + ".gfl_reg", feat, 4 * (reg_max + 1), 3)
    for l in range(n_levels):
        t["%s.scales.%d.scale" % (p, l)] = ()
    t[p + ".integral.project"] = (reg_max + 1,)


def mp_head_table(t: "_Table", p: str, proxies_list, in_channels: int = 256, feat: int = 256, stacked: int = 4,
                  reg_max: int = 16, n_levels: int = 5, num_words: int = 200):
    """mp_head.py:42-91: own parameter/buffers first, then the child modules."""
    nc = len(proxies_list)
    t[p + ".proxies"] = (sum(proxies_list), feat)
    t[p + "._embedding"] = (nc + 1, num_words, feat)
    t[p + "._pos_embedding_ptr"] = (nc + 1,)
    t[p + "._proxies_prob"] = (sum(proxies_list),)
    _gn_tower(t, p, in_channels, feat, stacked)
    t.plain(p + ".gfl_cls_conv", feat, feat, 3)
    t.plain(p + ".gfl_reg", feat, 4 * (reg_max + 1), 3)
    for l in range(n_levels):
        t["%s.scales.%d.scale" % (p, l)] = ()
    t[p + ".integral.project"] = (reg_max + 1,)


def resdet_state_dict_shapes(kind: str, num_classes: int = 10, depth: int = 50, start_level: int = 1,
                             num_outs: int = 5, add_extra_convs="on_output", stacked: int = 4, reg_max: int = 16,
                             proxies_list=(2, 3, 2, 5, 4, 8, 8, 4, 3, 3), gl_fusion: bool = False,
                             gl_levels=(1, 2, 3), gl_channel_cat: str = "linear") -> "OrderedDict[str, Tuple[int, ...]]":
    """kind 'gfl': GFL r50-FPN (SingleStageDetector: backbone / neck / bbox_head); 'mpdet': MPDet.
    gl_fusion: the neck is a GLFusionFPN (plug-ins neck.gl_fusion.<i> on the backbone outputs gl_levels)."""
    if kind not in ("gfl", "mpdet"):
        raise ValueError("kind must be 'gfl' or 'mpdet'")
    t = _Table()
    resnet_table(t, "backbone", depth)
    fpn_table(t, "neck", [256, 512, 1024, 2048], 256, start_level, num_outs, add_extra_convs)
    if gl_fusion:
        for i in gl_levels:
            gl_fusion_table(t, "neck.gl_fusion.%d" % i, [256, 512, 1024, 2048][i], gl_channel_cat)
    if kind == "gfl":
        gfl_head_table(t, "bbox_head", num_classes, 256, 256, stacked, reg_max, num_outs)
    else:
        mp_head_table(t, "bbox_head", proxies_list, 256, 256, stacked, reg_max, num_outs)
    return t
