"""glsdet_amd/compat: import-name shims (mmdet / mmcv / pycocotools / cv2) that let the reference's two-stage eval script
run as shipped (north_star: "ufpmp_det_eval.py run unchanged against the new backend").

CPU, only where the reference checkout is mounted (the build container): the script itself is imported with the shim
directory in front of sys.path -- every name it imports resolves, none of it is edited or copied.
GPU: the calls the script makes, in its order, on a synthetic frame written to disk (init_detector -> LoadImage-like
dict -> Compose(cfg.data.test.pipeline[1:]) -> collate -> scatter -> model(return_loss=False, rescale=True, **data) ->
UnifiedForegroundPacking -> cv2.imread / cv2.resize mosaic -> second detector -> COCO / loadRes / COCOeval), compared
with glsdet_amd's own two_stage_detect on the same frame."""
import importlib
import importlib.util
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "glsdet_amd", "compat")
REF_SCRIPTS = ["/root/reference/yolox-ufp/ufpmp_det_eval.py",
               "/root/reference/yolox-ufp/UFPMP-Det-Tools/eval_script/ufpmp_det_eval.py"]
SHIMMED = ("mmdet", "mmcv", "pycocotools", "cv2")


@pytest.fixture()
def shim_path():
    saved = {k: v for k, v in sys.modules.items() if k.split(".")[0] in SHIMMED}
    for k in saved:
        del sys.modules[k]
    sys.path.insert(0, SHIM)
    yield
    sys.path.remove(SHIM)
    for k in [k for k in sys.modules if k.split(".")[0] in SHIMMED]:
        del sys.modules[k]
    sys.modules.update(saved)


@pytest.mark.parametrize("path", REF_SCRIPTS)
def test_reference_eval_script_imports_unchanged_under_the_shim(shim_path, path):
    if not os.path.exists(path):
        pytest.skip("the reference checkout is not mounted here")
    spec = importlib.util.spec_from_file_location("ref_eval_under_shim", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)                       # module level: imports + defs; main() stays uncalled
    import glsdet_amd.eval
    import glsdet_amd.ufp
    assert mod.COCO is glsdet_amd.eval.COCO and mod.COCOeval is glsdet_amd.eval.COCOeval
    assert mod.UnifiedForegroundPacking is glsdet_amd.ufp.unified_foreground_packing
    assert mod.init_detector.__module__ == "mmdet.apis" and mod.Compose.__module__ == "mmdet.datasets.pipelines"
    assert callable(mod.collate) and callable(mod.scatter) and callable(mod.cv2.resize) and callable(mod.mmcv.imread)
    for fn in ("my_inference_detector", "display_merge_result", "py_cpu_nms", "compute_iof", "main", "LoadImage"):
        assert hasattr(mod, fn)


def test_compose_accepts_exactly_the_reference_test_pipeline(shim_path):
    from mmdet.datasets.pipelines import Compose
    from glsdet_amd.mmdet_surface import Config
    cfg = Config.fromfile(os.path.join(ROOT, "configs/UFPMP-Det/mp_det_res50.py"))
    c = Compose(cfg.data.test.pipeline[1:])
    assert c.args["img_scale"] == (1333, 800) and c.args["size_divisor"] == 32
    with pytest.raises(NotImplementedError):
        Compose([dict(type="MultiScaleFlipAug", img_scale=(1333, 800), flip=True, transforms=[])])


@pytest.mark.gpu
def test_the_scripts_call_sequence_through_the_shim(shim_path, tmp_path):
    import math
    import torch
    from PIL import Image
    from mmdet.apis import init_detector
    from mmdet.core import UnifiedForegroundPacking
    from mmdet.datasets.pipelines import Compose
    from mmcv.parallel import collate, scatter
    from pycocotools.coco import COCO
    from pycocotools.cocoeval import COCOeval
    import cv2
    import mmcv
    from glsdet_amd.synth import synth_input, synth_resdet_state_dict
    from tests.golden.make_golden import synth_image
    # synthetic "data set": one frame on disk, COCO annotations, two random checkpoints in mmcv's layout
    frame = synth_image((270, 480), 3)
    Image.fromarray(frame, "RGB").save(tmp_path / "f0.png")
    ann = dict(images=[dict(id=0, width=480, height=270, file_name="f0.png")], categories=[dict(id=c) for c in range(10)],
               annotations=[dict(id=1, image_id=0, category_id=3, bbox=[40, 50, 60, 40], area=2400, iscrowd=0)])
    (tmp_path / "ann.json").write_text(json.dumps(ann))
    cal = synth_input((1, 3, 128, 160), 100)
    for kind, name in (("gfl", "coarse.pth"), ("mpdet", "fine.pth")):
        torch.save({"state_dict": synth_resdet_state_dict(kind, 0, cal), "meta": {}}, tmp_path / name)
    device = "cuda"
    coarse = init_detector(os.path.join(ROOT, "configs/UFPMP-Det/coarse_det.py"), str(tmp_path / "coarse.pth"), device=device)
    fine = init_detector(os.path.join(ROOT, "configs/UFPMP-Det/mp_det_res50.py"), str(tmp_path / "fine.pth"), device=device)
    assert next(coarse.parameters()).is_cuda
    coco = COCO(str(tmp_path / "ann.json"))
    width, height = coco.imgs[0]["width"], coco.imgs[0]["height"]
    img_path = str(tmp_path / coco.imgs[0]["file_name"])

    def infer(model, img):                              # ufpmp_det_eval.py:107-146 (LoadImage + my_inference_detector)
        data = dict(img=img, img_fields=["img"], img_shape=img.shape, ori_shape=img.shape, filename=None, ori_filename=None)
        data = Compose(model.cfg.data.test.pipeline[1:])(data)
        data = scatter(collate([data], samples_per_gpu=1), [next(model.parameters()).device])[0]
        with torch.no_grad():
            return model(return_loss=False, rescale=True, **data)[0]

    def set_thr(model, img, keep):
        """random-init heads fire everywhere: raise test_cfg.score_thr so that ~keep (position, class) pairs pass"""
        data = Compose(model.cfg.data.test.pipeline[1:])(dict(img=img))
        cls, _ = model._detector().forward_raw(data["img"][0][None])
        p = torch.sigmoid(torch.cat([c.flatten() for c in cls]))
        model.bbox_head.test_cfg["score_thr"] = float(torch.topk(p, keep).values[-1])
        return model.bbox_head.test_cfg["score_thr"]
    t1 = set_thr(coarse, mmcv.imread(img_path), 60)
    first = infer(coarse, mmcv.imread(img_path))
    assert len(first) == 10 and all(r.shape[1] == 5 for r in first)
    boxes = np.concatenate(first)
    if len(boxes) == 0:
        pytest.skip("the random coarse detector found nothing on this frame")
    rec, w, h = UnifiedForegroundPacking(boxes[:, :4], 1.5, input_shape=[width, height])
    img_data = cv2.imread(img_path)
    assert img_data.dtype == np.uint8 and img_data.shape == (270, 480, 3) and np.array_equal(img_data[:, :, ::-1], frame)
    canvas = np.zeros((math.ceil(h), math.ceil(w), 3))
    for chip in rec:                                    # display_merge_result, :182-193
        x1, y1, cw, ch, nx, ny, s = [math.floor(v) for v in chip]
        if cw == 0 or ch == 0:
            continue
        canvas[ny:ny + ch * s, nx:nx + cw * s, :] = cv2.resize(img_data[y1:y1 + ch, x1:x1 + cw, :], (cw * s, ch * s))
    t2 = set_thr(fine, canvas, 400)
    second = infer(fine, canvas)
    assert len(second) == 10
    # the same frame through glsdet_amd's own two-stage driver: same mosaic, same fine detections
    from glsdet_amd.ufp import UfpSecondStage, two_stage_detect
    _, info = two_stage_detect(coarse._detector(), fine._detector(), img_data, UfpSecondStage(),
                               dict(score_thr=t1, iou_thr=0.6, nms_pre=1000, max_per_img=100),
                               dict(score_thr=t2, iou_thr=0.6, nms_pre=1000, max_per_img=500))
    assert np.array_equal(info["canvas"].cpu().numpy(), canvas.astype(np.float32))
    # COCO hand-off (:326-338)
    res = [dict(image_id=0, category_id=c, score=float(r[4]), bbox=[int(r[0]), int(r[1]), int(r[2]) - int(r[0]), int(r[3]) - int(r[1])])
           for c, rows in enumerate(second) for r in rows][:50] or [dict(image_id=0, category_id=3, score=0.5, bbox=[40, 50, 60, 40])]
    (tmp_path / "res.json").write_text(json.dumps(res))
    E = COCOeval(coco, coco.loadRes(str(tmp_path / "res.json")), "bbox")
    E.params.maxDets = [10, 100, 500]
    E.evaluate(); E.accumulate(); E.summarize()
    assert len(E.stats) == 12
