"""`from mmdet.apis import init_detector, show_result_pyplot, inference_detector` (ufpmp_det_eval.py:2)."""
import torch

from glsdet_amd.mmdet_surface import init_detector as _init


def init_detector(config, checkpoint=None, device="cuda:0", cfg_options=None):
    """ufp/mmdet/apis/inference.py:17-53.  The parameter container follows `.to(device)` so that
    `next(model.parameters()).is_cuda` (ufpmp_det_eval.py:127) sees the device the HIP plan runs on."""
    model = _init(config, checkpoint, device=device, cfg_options=cfg_options)
    model.to(torch.device(device))
    return model


def inference_detector(model, imgs):
    """ufp/mmdet/apis/inference.py:82-151 for ndarray / path inputs, through the same pipeline shim."""
    from mmdet.datasets.pipelines import Compose
    from mmcv.parallel import collate, scatter
    import mmcv
    single = not isinstance(imgs, (list, tuple))
    out = []
    for im in ([imgs] if single else imgs):
        img = mmcv.imread(im) if isinstance(im, str) else im
        data = dict(img=img, img_fields=["img"], img_shape=img.shape, ori_shape=img.shape, filename=None, ori_filename=None)
        data = Compose(model.cfg.data.test.pipeline[1:])(data)
        data = scatter(collate([data], samples_per_gpu=1), [next(model.parameters()).device])[0]
        with torch.no_grad():
            out.append(model(return_loss=False, rescale=True, **data)[0])
    return out[0] if single else out


def show_result_pyplot(*args, **kwargs):
    raise NotImplementedError("visualisation is outside the detection forward path")
