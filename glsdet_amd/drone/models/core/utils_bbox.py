"""Twin of drone/models/core/utils_bbox.py (the functions the reference harness calls,
yolo.py:143-150): same names, arguments and return values; arithmetic in libglsdet_hip."""
from __future__ import annotations

import numpy as np
import torch

from glsdet_amd._lib import F32
from glsdet_amd.engine import Engine

_ENG = {}


def _engine(device="cuda:0"):
    if device not in _ENG:
        _ENG[device] = Engine("f32", device)
    return _ENG[device]


def yolo_correct_boxes(box_xy, box_wh, input_shape, image_shape, letterbox_image):
    """drone/models/core/utils_bbox.py:8-33 -- numpy on the few kept boxes, as in the
    reference (it runs after `.cpu().numpy()` there as well, :481-483).  Returns
    [y1, x1, y2, x2] in original-image pixels."""
    box_yx = box_xy[..., ::-1]
    box_hw = box_wh[..., ::-1]
    input_shape = np.array(input_shape)
    image_shape = np.array(image_shape)
    if letterbox_image:
        new_shape = np.round(image_shape * np.min(input_shape / image_shape))
        offset = (input_shape - new_shape) / 2. / input_shape
        scale = input_shape / new_shape
        box_yx = (box_yx - offset) * scale
        box_hw = box_hw * scale
    box_mins = box_yx - (box_hw / 2.)
    box_maxes = box_yx + (box_hw / 2.)
    boxes = np.concatenate([box_mins[..., 0:1], box_mins[..., 1:2], box_maxes[..., 0:1], box_maxes[..., 1:2]], axis=-1)
    boxes *= np.concatenate([image_shape, image_shape], axis=-1)
    return boxes


def decode_outputs(outputs, input_shape):
    """[B,5+nc,H,W] x levels -> [B, A, 5+nc] (sigmoid on obj/cls, grid decode, normalised by
    input (w, h)); drone/models/core/utils_bbox.py:254-306.  Unlike the reference it does not
    mutate `outputs`."""
    comp = getattr(outputs, "compiled", None)
    nc = outputs[0].shape[1] - 5
    H, W = int(input_shape[0]), int(input_shape[1])
    if comp is not None:                       # native fp32 NHWC levels of our own forward
        return comp.eng.decode(comp.levels, nc, H, W, mode=0)
    eng = _engine()
    levels = []
    for o in outputs:
        n, c, h, w = o.shape
        v = eng.tensor(n, h, w, c, F32)
        dst = torch.as_strided(v.buf.view(torch.float32), (n, h, w, v.c), (v.sn, v.sh, v.sw, 1))
        dst[..., :c] = o.detach().to(eng.device, torch.float32).permute(0, 2, 3, 1)
        levels.append(v)
    return eng.decode(levels, nc, H, W, mode=0)


def non_max_suppression(prediction, num_classes, input_shape, image_shape, letterbox_image, conf_thres=0.5,
                        nms_thres=0.4):
    """drone/models/core/utils_bbox.py:375-484.  Per image: None when nothing passes, else
    ndarray(n, 7) = [y1, x1, y2, x2, obj_conf, class_conf, class_pred] in original-image pixels."""
    eng = _engine(str(prediction.device) if prediction.is_cuda else "cuda:0")
    pred = prediction.detach().to(eng.device, torch.float32).contiguous()
    n, A = pred.shape[0], pred.shape[1]
    if A == 0:
        return [None for _ in range(n)]
    nb = eng.nms_buffers(n, A, A, A)
    dets, count, status = eng.nms(pred, num_classes, 0, float(conf_thres), float(nms_thres), nb)
    count = count.cpu().numpy()
    if int(status.item()) & 1:
        raise RuntimeError("NMS candidate capacity exceeded")
    dets = dets.cpu().numpy()
    output = [None for _ in range(n)]
    for i in range(n):
        if count[i] == 0:
            continue
        d = dets[i, : count[i]].copy()
        box_xy, box_wh = (d[:, 0:2] + d[:, 2:4]) / 2, d[:, 2:4] - d[:, 0:2]
        d[:, :4] = yolo_correct_boxes(box_xy, box_wh, input_shape, image_shape, letterbox_image)
        output[i] = d
    return output
