#!/usr/bin/env python3
"""bbox COCOeval at VisDrone-val size (548 images, 10 categories, ~70 ground truths and up to 500 detections
per image, maxDets [10,100,500]): the device matching (`glsdet_coco_match`) against the oracle's Python
loops on a sample of the pairs.  Run on the GPU box."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def visdrone_like(n_img=548, n_cat=10, seed=0):
    r = np.random.default_rng(seed)
    anns, res = [], []
    for im in range(n_img):
        for c in range(n_cat):
            ng = int(r.poisson(7))
            xy = r.uniform(0, 1800, (ng, 2)); wh = r.uniform(6, 120, (ng, 2))
            for k in range(ng):
                bb = [float(xy[k, 0]), float(xy[k, 1]), float(wh[k, 0]), float(wh[k, 1])]
                anns.append(dict(id=len(anns) + 1, image_id=im, category_id=c, bbox=bb, area=bb[2] * bb[3], iscrowd=int(r.random() < 0.02)))
                if r.random() < 0.8:
                    j = r.normal(0, 2.0, 4)
                    res.append(dict(image_id=im, category_id=c, score=float(r.uniform(0.3, 1)),
                                    bbox=[bb[0] + j[0], bb[1] + j[1], max(2.0, bb[2] + j[2]), max(2.0, bb[3] + j[3])]))
            for _ in range(int(r.poisson(40))):
                res.append(dict(image_id=im, category_id=c, score=float(r.uniform(0.05, 0.6)),
                                bbox=[float(v) for v in r.uniform(0, 1800, 2)] + [float(v) for v in r.uniform(6, 120, 2)]))
    ds = dict(images=[dict(id=i) for i in range(n_img)], categories=[dict(id=c) for c in range(n_cat)], annotations=anns)
    return ds, res


def main():
    import torch
    from glsdet_amd.eval import COCO, COCOeval
    from oracle import cocoeval_oracle as CO
    ds, res = visdrone_like()
    print("ground truths %d, detections %d" % (len(ds["annotations"]), len(res)), flush=True)
    gt = COCO(ds)
    dt = gt.loadRes(res)
    for rep in range(2):
        E = COCOeval(gt, dt, "bbox")
        E.params.maxDets = [10, 100, 500]
        t0 = time.time(); E.evaluate(); t1 = time.time(); E.accumulate(); t2 = time.time()
        # the device part alone
        print("evaluate %.2f s (pack + device + unpack)   accumulate %.2f s" % (t1 - t0, t2 - t1), flush=True)
    E.summarize()
    # device time of the three kernels
    import ctypes
    st = torch.cuda.Event(enable_timing=True); en = torch.cuda.Event(enable_timing=True)
    orig = E._match
    def timed(*a):
        st.record(); out = orig(*a); en.record(); torch.cuda.synchronize()
        print("glsdet_coco_match + copies: %.2f ms" % st.elapsed_time(en), flush=True)
        return out
    E._match = timed
    E.evaluate()
    # oracle on a sample of images
    n = 12
    sub = dict(images=ds["images"][:n], categories=ds["categories"], annotations=[a for a in ds["annotations"] if a["image_id"] < n])
    sres = [r for r in res if r["image_id"] < n]
    t0 = time.time(); stats, ev, imgs, pe = CO.coco_eval(sub, sres, max_dets=(10, 100, 500)); t = time.time() - t0
    print("oracle (python loops, 1 core): %d images in %.1f s -> %.1f s for 548" % (n, t, t * 548 / n), flush=True)
    Es = COCOeval(COCO(sub), COCO(sub).loadRes(sres), "bbox"); Es.params.maxDets = [10, 100, 500]
    Es.evaluate(); Es.accumulate(); Es.summarize()
    print("sample stats identical to the oracle:", bool(np.array_equal(Es.stats, stats)))


if __name__ == "__main__":
    main()
