# Coarse detector of UFPMP-Det (ufp/ufpmp_det_eval.py:218): GFL, ResNet-50 + FPN, 10 VisDrone
# classes.  The reference checkout names this file but does not ship it; the keys below are the
# ones its registered classes take (backbones/resnet.py:371, necks/fpn.py:62,
# dense_heads/gfl_head.py:87) with the values of the GFL r50-FPN recipe.
_base_ = ['../_base_/default_runtime.py']
model = dict(
    type='GFL',
    backbone=dict(
        type='ResNet', depth=50, num_stages=4, out_indices=(0, 1, 2, 3), frozen_stages=1,
        norm_cfg=dict(type='BN', requires_grad=True), norm_eval=True, style='pytorch'),
    neck=dict(
        type='FPN', in_channels=[256, 512, 1024, 2048], out_channels=256, start_level=1,
        add_extra_convs='on_output', num_outs=5),
    bbox_head=dict(
        type='GFLHead', num_classes=10, in_channels=256, stacked_convs=4, feat_channels=256,
        anchor_generator=dict(type='AnchorGenerator', ratios=[1.0], octave_base_scale=8, scales_per_octave=1,
                              strides=[8, 16, 32, 64, 128]),
        loss_cls=dict(type='QualityFocalLoss', use_sigmoid=True, beta=2.0, loss_weight=1.0),
        loss_dfl=dict(type='DistributionFocalLoss', loss_weight=0.25),
        reg_max=16,
        loss_bbox=dict(type='GIoULoss', loss_weight=2.0)),
    train_cfg=dict(assigner=dict(type='ATSSAssigner', topk=9), allowed_border=-1, pos_weight=-1, debug=False),
    test_cfg=dict(nms_pre=1000, min_bbox_size=0, score_thr=0.05, nms=dict(type='nms', iou_threshold=0.6),
                  max_per_img=100))
img_norm_cfg = dict(mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_rgb=True)
test_pipeline = [
    dict(type='LoadImageFromFile'),
    dict(type='MultiScaleFlipAug', img_scale=(1333, 800), flip=False,
         transforms=[dict(type='Resize', keep_ratio=True), dict(type='RandomFlip'),
                     dict(type='Normalize', **img_norm_cfg), dict(type='Pad', size_divisor=32),
                     dict(type='ImageToTensor', keys=['img']), dict(type='Collect', keys=['img'])])]
data = dict(samples_per_gpu=8, workers_per_gpu=2, test=dict(type='CocoDataset', pipeline=test_pipeline))
