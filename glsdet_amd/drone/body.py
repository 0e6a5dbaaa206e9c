"""`YoloBody(num_classes, phi)` twin of the reference's detector modules
(drone/models/base/yolox.py:237-251, block/non_local/yolo_patch_nonlocal_plus.py:249-263).

It is an nn.Module only as a parameter container with the reference's exact state_dict
(names, shapes, registration order); `forward` runs the recorded libglsdet_hip plan.
There is no PyTorch/CPU execution path: without the HIP library or a GPU it raises.
"""
from __future__ import annotations

from typing import List

import torch
import torch.nn as nn

from ..arch import state_dict_shapes
from ..detector import HipDetector


def _autotune() -> bool:
    """Kernel variants are measured per conv problem when a plan is first built for an input shape (a few
    seconds, +6 % throughput); GLSDET_AUTOTUNE=0 switches to the size heuristic."""
    import os
    return os.environ.get("GLSDET_AUTOTUNE", "1") != "0"


class RawOutputs(list):
    """list of [B, 5+nc, H_l, W_l] logits (what the reference returns) that also keeps the
    native NHWC fp32 level views, so `decode_outputs` can consume them without a round trip."""
    compiled = None
    detector = None


class TableModule(nn.Module):
    """nn.Module used purely as a parameter container: registers the tensors of a
    {key: shape} table (reference names, reference order) and exposes them under exactly those
    names in state_dict()/load_state_dict(); arithmetic happens elsewhere (libglsdet_hip)."""

    def _init_table(self, shapes):
        self._names = {}
        for key, shape in shapes.items():
            flat = key.replace(".", "__")
            self._names[flat] = key
            if key.endswith("num_batches_tracked"):
                self.register_buffer(flat, torch.zeros(shape, dtype=torch.long))
            elif key.endswith("running_mean"):
                self.register_buffer(flat, torch.zeros(shape))
            elif key.endswith("running_var"):
                self.register_buffer(flat, torch.ones(shape))
            elif key.endswith(("_embedding", "_proxies_prob")):          # MPHead buffers (mp_head.py:78-91)
                self.register_buffer(flat, torch.randn(shape) if key.endswith("_embedding") else torch.ones(shape))
            elif key.endswith("_pos_embedding_ptr"):
                self.register_buffer(flat, torch.zeros(shape, dtype=torch.long))
            elif key.endswith("integral.project"):                        # gfl_head.py:32-33
                self.register_buffer(flat, torch.linspace(0, shape[0] - 1, shape[0]))
            elif key.endswith("proxies"):
                self.register_parameter(flat, nn.Parameter(torch.randn(shape) * 0.01, requires_grad=False))
            elif (key.endswith("weight") and len(shape) == 1) or len(shape) == 0:   # BN / GN gamma, mmcv Scale
                self.register_parameter(flat, nn.Parameter(torch.ones(shape), requires_grad=False))
            elif key.endswith("bias"):
                self.register_parameter(flat, nn.Parameter(torch.zeros(shape), requires_grad=False))
            else:       # conv weights: torch's default kaiming-uniform(a=sqrt(5)) bound
                fan_in = shape[1] * shape[2] * shape[3]
                bound = 1.0 / fan_in ** 0.5
                self.register_parameter(flat, nn.Parameter(torch.empty(shape).uniform_(-bound, bound),
                                                           requires_grad=False))
        self._on_weights_changed()

    def _on_weights_changed(self):
        pass

    # ---- reference-named state_dict -----------------------------------------------------
    def state_dict(self, *args, destination=None, prefix="", keep_vars=False):
        raw = nn.Module.state_dict(self, prefix="", keep_vars=keep_vars)
        out = destination if destination is not None else type(raw)()
        for flat, key in self._names.items():          # the reference's registration order
            out[prefix + key] = raw[flat]
        return out

    def load_state_dict(self, state_dict, strict: bool = True):
        inv = {v: k for k, v in self._names.items()}
        sd, unexpected = {}, []
        for k, v in state_dict.items():
            k = k[7:] if k.startswith("module.") else k          # DataParallel checkpoints
            if k in inv:
                sd[inv[k]] = v
            else:
                unexpected.append(k)
        missing = [self._names[f] for f in self._names if f not in sd]
        if strict and (missing or unexpected):
            raise RuntimeError("Error(s) in loading state_dict for {}: missing {} unexpected {}".format(
                type(self).__name__, missing[:5], unexpected[:5]))
        res = nn.Module.load_state_dict(self, sd, strict=False)
        self._on_weights_changed()                                # weights changed: re-pack lazily
        return res


class HipYoloBody(TableModule):
    kind = "base"
    attention_backbone = False          # True: new/darknet_att.py, "lsk": lsk/darknet_lsk.py

    def __init__(self, num_classes: int, phi: str, dtype: str = "f16"):
        super().__init__()
        self.num_classes, self.phi, self.hip_dtype = num_classes, phi, dtype
        self._det = None
        self._init_table(state_dict_shapes(self.kind, phi, num_classes, self.attention_backbone))

    def _on_weights_changed(self):
        self._det = None

    # ---- forward ---------------------------------------------------------------------------
    def forward(self, x: torch.Tensor) -> List[torch.Tensor]:
        if self.training:
            raise NotImplementedError("glsdet_amd implements the inference forward only: call .eval()")
        if self._det is None:
            self._det = HipDetector(self.kind, self.state_dict(), dtype=self.hip_dtype, autotune=_autotune())
        det = self._det
        x = x.to("cuda", torch.float32)
        n, _, H, W = x.shape
        c = det.compile(n, H, W)
        det.run(c, x)
        out = RawOutputs(l.to_nchw(5 + det.num_classes) for l in c.levels)
        out.compiled, out.detector = c, det
        return out


class BaseYoloBody(HipYoloBody):
    kind = "base"


class GLYoloBody(HipYoloBody):
    kind = "gl"


class CrossYoloBody(HipYoloBody):
    """YOLOX with the cross-scale decoupled head (drone/models/new/yolox6.py, whose import of
    `models.decouple` is missing from the reference checkout; head text-identical to
    drone/models/lsk/yolox6.py) on the plain CSPDarknet backbone."""
    kind = "cross"


class LskCrossYoloBody(CrossYoloBody):
    """drone/models/lsk/yolox6.py = lsk/yolox6_lsk.py: the cross-scale head on lsk/darknet_lsk.py's backbone (an
    LSK.Attention block -- 1x1, GELU, LSKblock, 1x1 + shortcut -- after every stage)."""
    attention_backbone = "lsk"
