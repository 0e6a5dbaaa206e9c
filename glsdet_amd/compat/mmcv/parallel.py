"""`from mmcv.parallel import collate, scatter` (ufpmp_det_eval.py:8): batch of one sample, already on the device."""
import torch


def collate(batch, samples_per_gpu=1):
    """[dict(img=[Tensor[3,H,W]], img_metas=[meta])] -> dict(img=[Tensor[B,3,H,W]], img_metas=[[meta, ...]])"""
    if samples_per_gpu != len(batch) and samples_per_gpu != 1:
        raise NotImplementedError("collate: one device batch")
    n_aug = len(batch[0]["img"])
    return dict(img=[torch.stack([b["img"][a] for b in batch]) for a in range(n_aug)],
                img_metas=[[b["img_metas"][a] for b in batch] for a in range(n_aug)])


def scatter(inputs, target_gpus, dim=0):
    if len(target_gpus) != 1:
        raise NotImplementedError("scatter: one process per GPU")
    dev = target_gpus[0] if isinstance(target_gpus[0], torch.device) else torch.device("cuda", int(target_gpus[0]))
    return [dict(img=[t.to(dev) for t in inputs["img"]], img_metas=inputs["img_metas"])]
