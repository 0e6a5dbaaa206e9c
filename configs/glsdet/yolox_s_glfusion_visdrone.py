# GLSDet: YOLOX-s with the Global-Local fusion neck behind the mmdet config surface.
_base_ = ['../yolox/yolox_s_visdrone.py']
model = dict(neck=dict(type='GLFusionPAFPN'))
