"""String-keyed plug-in factories: the seam the HIP path hides behind.

Mirrors models/registry.py of the reference: same function names, same cfg keys read,
the same four cfg keys written in place ('model.backbone.out_channels'/'out_strides',
'model.fpn.out_channels'/'out_strides'), same exceptions on unknown names.
Pretrained files the reference hard-requires (weights/dark53_imgnet.pth,
models/registry.py:15) are loaded when present and skipped with a notice otherwise.
"""
import os

import torch


def get_backbone(cfg: dict):
    '''
    Get backbone network (reference: models/registry.py:4-40)
    '''
    backbone_name = cfg['model.backbone.name']
    if backbone_name == 'dark53':
        from .backbones import Darknet53
        from .. import PROJECT_ROOT
        assert cfg['model.backbone.num_levels'] == 3
        backbone = Darknet53(cfg)
        path = f'{PROJECT_ROOT}/weights/dark53_imgnet.pth'
        if os.path.exists(path):
            print("Using backbone Darknet-53. Loading ImageNet weights....")
            backbone.load_state_dict(torch.load(path), strict=True)
        else:
            print(f"Using backbone Darknet-53. No ImageNet weights at {path}; keeping initial weights.")
        out_feature_channels = (256, 512, 1024)
        out_strides = (8, 16, 32)
    elif backbone_name.startswith('efficientnet'):
        from .backbones import EfNetBackbone
        backbone = EfNetBackbone(cfg)            # the reference downloads ImageNet weights here; none offline
        out_feature_channels = backbone.feature_chs
        out_strides = backbone.feature_strides
    else:
        raise Exception('Unknown backbone name')

    cfg['model.backbone.out_channels'] = out_feature_channels
    cfg['model.backbone.out_strides'] = out_strides
    return backbone


def get_fpn(cfg: dict):
    '''
    Get feature pyramid network (reference: models/registry.py:43-75)
    '''
    fpn_name = cfg['model.fpn.name']
    if fpn_name == 'yolov3':
        from .fpns import YOLOv3FPN
        fpn = YOLOv3FPN(cfg)
        out_feature_channels = cfg['model.backbone.out_channels']
        out_strides = cfg['model.backbone.out_strides']
    elif fpn_name == 'bifpn':
        from .fpns import get_bifpn
        fpn = get_bifpn(cfg)
        ch = cfg['model.bifpn.out_ch']
        out_feature_channels = [ch for _ in cfg['model.backbone.out_channels']]
        out_strides = cfg['model.backbone.out_strides']
    else:
        raise Exception('Unknown FPN name')

    cfg['model.fpn.out_channels'] = out_feature_channels
    cfg['model.fpn.out_strides'] = out_strides
    return fpn


def get_rpn(cfg: dict):
    '''
    Get the detection head (reference: models/registry.py:100-116)
    '''
    rpn_name = cfg['model.rpn.name']
    if rpn_name == 'yolov3':
        from .rpns import YOLOHead
        rpn = YOLOHead(cfg)
    elif rpn_name == 'effrpn':
        from .rpns import EfDetHead
        rpn = EfDetHead(cfg)
    elif rpn_name == 'effrpn_ct':
        from .rpns import EfDetHead_wCenter
        rpn = EfDetHead_wCenter(cfg)
    else:
        raise NotImplementedError()
    return rpn


def get_det_layer(cfg: dict):
    '''
    Get the final decode layer CLASS (reference: models/registry.py:119-146)
    '''
    det_layer_name = cfg['model.pred_layer']
    if det_layer_name == 'YOLO':
        from .detlayers.yolov3 import YOLOLayer
        return YOLOLayer
    elif det_layer_name == 'RetinaNet':
        from .detlayers.retinanet import RetinaLayer
        return RetinaLayer
    elif det_layer_name == 'FCOS':
        from .detlayers.fcos import FCOSLayer
        return FCOSLayer
    elif det_layer_name == 'FCOS2':
        from .detlayers.fcos2 import FCOSLayer
        return FCOSLayer
    elif det_layer_name == 'FCOS2_ATSS':
        from .detlayers.fcos2 import FCOS_ATSS_Layer
        return FCOS_ATSS_Layer
    else:
        raise NotImplementedError()
