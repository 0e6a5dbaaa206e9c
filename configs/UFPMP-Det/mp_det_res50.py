# Fine detector of UFPMP-Det (ufp/ufpmp_det_eval.py:220): MPDet = ResNet-50 + FPN + MPHead
# (detectors/mpdet.py:9-18, dense_heads/mp_head.py:23-39), 10 VisDrone classes, 42 proxies.
_base_ = ['./coarse_det.py']
model = dict(
    type='MPDet',
    bbox_head=dict(
        _delete_=True,
        type='MPHead', num_classes=10, in_channels=256, stacked_convs=4, feat_channels=256,
        num_words=200, beta=0, gamma=10, proxies_list=[2, 3, 2, 5, 4, 8, 8, 4, 3, 3],
        anchor_generator=dict(type='AnchorGenerator', ratios=[1.0], octave_base_scale=8, scales_per_octave=1,
                              strides=[8, 16, 32, 64, 128]),
        loss_cls=dict(type='QualityFocalLoss', use_sigmoid=True, beta=2.0, loss_weight=1.0),
        loss_dfl=dict(type='DistributionFocalLoss', loss_weight=0.25),
        reg_max=16,
        loss_bbox=dict(type='GIoULoss', loss_weight=2.0)),
    test_cfg=dict(nms_pre=1000, min_bbox_size=0, score_thr=0.05, nms=dict(type='nms', iou_threshold=0.6),
                  max_per_img=500))
