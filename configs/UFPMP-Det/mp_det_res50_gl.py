# BASELINE config 3 as BASELINE.json names it: "mp_det_res50.py (ResNet-50 + GL-fusion + decoupled head), 1333x800".
# The reference ships neither this file nor such a model; this build authors it on the registry surface
# the reference does have: MPDet (detectors/mpdet.py:9-18) + ResNet-50 + MPHead (the decoupled cls / reg towers of
# dense_heads/mp_head.py:42-91), with the GL-fusion block wired as SURVEY App. B proposes -- the residual
# feat + Patch_Conv_NonLocal_new(feat) of drone/models/new/yolox10.py:262-266 on C3, C4, C5 before the FPN laterals
# (neck type GLFusionFPN, glsdet_amd/mmdet_surface/resdet_models.py).
_base_ = ['./mp_det_res50.py']
model = dict(
    neck=dict(
        _delete_=True,
        type='GLFusionFPN', in_channels=[256, 512, 1024, 2048], out_channels=256, start_level=1,
        add_extra_convs='on_output', num_outs=5,
        gl_levels=[1, 2, 3], gl_channel_cat='linear'))
