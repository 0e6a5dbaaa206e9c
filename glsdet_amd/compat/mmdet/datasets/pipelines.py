"""`from mmdet.datasets.pipelines import Compose` (ufpmp_det_eval.py:9).

`Compose(cfg.data.test.pipeline[1:])` of the reference configs is MultiScaleFlipAug(img_scale, flip=False,
transforms=[Resize(keep_ratio), RandomFlip, Normalize, Pad(size_divisor), ImageToTensor, Collect])
(ufp/configs/_base_/datasets/coco_detection.py:16-30; mmdet/datasets/pipelines/{transforms,test_time_aug,formating}.py).
Here the whole chain is ONE device call (glsdet_amd.ufp.UfpSecondStage.pipeline_input: uint8 frame -> cv2's fixed-point
bilinear resize; float mosaic -> float bilinear; normalise; pad): the result dict has the keys Collect emits."""
import numpy as np
import torch


class Compose:
    def __init__(self, transforms):
        self.transforms = list(transforms)
        msfa = [t for t in self.transforms if t["type"] == "MultiScaleFlipAug"]
        if len(msfa) != 1 or len(self.transforms) != 1:
            raise NotImplementedError("the test pipeline must be a single MultiScaleFlipAug (as the reference configs build it)")
        t = msfa[0]
        scale = t["img_scale"]
        if isinstance(scale, list):
            if len(scale) != 1:
                raise NotImplementedError("multi-scale test-time augmentation is outside the hot path")
            scale = scale[0]
        if t.get("flip", False):
            raise NotImplementedError("flip test-time augmentation is outside the hot path")
        inner = {s["type"]: s for s in t["transforms"]}
        unknown = set(inner) - {"Resize", "RandomFlip", "Normalize", "Pad", "ImageToTensor", "Collect", "DefaultFormatBundle"}
        if unknown or not inner.get("Resize", {}).get("keep_ratio", False) or "Normalize" not in inner:
            raise NotImplementedError("pipeline steps %s are not lowered" % sorted(unknown or inner))
        n = inner["Normalize"]
        if not n.get("to_rgb", True):
            raise NotImplementedError("Normalize(to_rgb=False) is not lowered")
        self.args = dict(img_scale=(int(max(scale)), int(min(scale))), size_divisor=int(inner.get("Pad", {}).get("size_divisor", 1)),
                         mean_rgb=tuple(n["mean"]), std_rgb=tuple(n["std"]))
        self._stage = None

    def __call__(self, results):
        from glsdet_amd.ufp import UfpSecondStage
        if self._stage is None:
            self._stage = UfpSecondStage(**self.args)
        img = results["img"]
        if isinstance(img, np.ndarray):
            # a decoded frame is uint8; the mosaic of display_merge_result is a float64 array of uint8-valued pixels
            t = torch.from_numpy(np.ascontiguousarray(img if img.dtype == np.uint8 else img.astype(np.float32)))
        else:
            t = img
        x, meta = self._stage.pipeline_input(t.to(self._stage.device))
        meta.update(filename=results.get("filename"), ori_filename=results.get("ori_filename"),
                    img_norm_cfg=dict(mean=np.array(self.args["mean_rgb"], np.float32), std=np.array(self.args["std_rgb"], np.float32),
                                      to_rgb=True))
        return dict(img=[x[0]], img_metas=[meta])
