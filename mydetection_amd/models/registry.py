"""String-keyed plug-in factories: the seam the HIP path hides behind.

Same four entry points as the reference's models/registry.py -- `get_backbone`, `get_fpn`, `get_rpn`,
`get_det_layer` --, reading the same cfg keys, writing the same four derived keys in place
('model.backbone.out_channels' / 'out_strides', 'model.fpn.out_channels' / 'out_strides') and raising the same
exceptions for names they do not know (bare `Exception('Unknown ...')` for backbone / FPN, `NotImplementedError`
for head / decode layer; models/registry.py:36,71,115,146).  Dispatch is by table: name -> (module, attribute),
imported on first use.
"""
import importlib
import os

import torch

_HEADS = {                                   # cfg['model.rpn.name']            reference: models/registry.py:100-116
    'yolov3': ('rpns', 'YOLOHead'),
    'effrpn': ('rpns', 'EfDetHead'),
    'effrpn_ct': ('rpns', 'EfDetHead_wCenter'),
}
_DECODE_LAYERS = {                           # cfg['model.pred_layer']          reference: models/registry.py:119-146
    'YOLO': ('detlayers.yolov3', 'YOLOLayer'),
    'RetinaNet': ('detlayers.retinanet', 'RetinaLayer'),
    'FCOS': ('detlayers.fcos', 'FCOSLayer'),
    'FCOS2': ('detlayers.fcos2', 'FCOSLayer'),
    'FCOS2_ATSS': ('detlayers.fcos2', 'FCOS_ATSS_Layer'),
}


def _resolve(entry):
    module, attr = entry
    return getattr(importlib.import_module(f'{__package__}.{module}'), attr)


def _darknet53(cfg):
    """Darknet-53 with the three taps of the reference (models/registry.py:10-24).  The ImageNet checkpoint the
    reference hard-requires (weights/dark53_imgnet.pth, :15) is loaded when present and skipped with a notice
    otherwise -- it does not exist offline."""
    from .. import PROJECT_ROOT
    assert cfg['model.backbone.num_levels'] == 3
    net = _resolve(('backbones', 'Darknet53'))(cfg)
    path = f'{PROJECT_ROOT}/weights/dark53_imgnet.pth'
    if os.path.exists(path):
        print("Using backbone Darknet-53. Loading ImageNet weights....")
        net.load_state_dict(torch.load(path), strict=True)
    else:
        print(f"Using backbone Darknet-53. No ImageNet weights at {path}; keeping initial weights.")
    return net, (256, 512, 1024), (8, 16, 32)


def _efficientnet(cfg):
    """EfficientNet-B* with the C1..C5 taps (+ C6/C7); the reference downloads ImageNet weights here
    (models/registry.py:25-34), which is not possible offline."""
    net = _resolve(('backbones', 'EfNetBackbone'))(cfg)
    return net, net.feature_chs, net.feature_strides


def _ultralytics(cfg):
    """YOLOv5 trunk (models/registry.py:25-28)."""
    net = _resolve(('backbones', 'UltralyticsBackbone'))(cfg)
    return net, net.feature_chs, net.feature_strides


def get_backbone(cfg: dict):
    '''
    Backbone network for cfg['model.backbone.name']; records its output channels and strides in cfg
    (reference: models/registry.py:4-40)
    '''
    name = cfg['model.backbone.name']
    build = (_darknet53 if name == 'dark53' else _ultralytics if name == 'ultralytics'
             else _efficientnet if name.startswith('efficientnet') else None)
    if build is None:
        raise Exception('Unknown backbone name')
    backbone, cfg['model.backbone.out_channels'], cfg['model.backbone.out_strides'] = build(cfg)
    return backbone


def get_fpn(cfg: dict):
    '''
    Feature pyramid for cfg['model.fpn.name']; records the pyramid's channels and strides in cfg
    (reference: models/registry.py:43-75)
    '''
    name = cfg['model.fpn.name']
    strides = cfg['model.backbone.out_strides']
    if name == 'yolov3':
        fpn, channels = _resolve(('fpns', 'YOLOv3FPN'))(cfg), cfg['model.backbone.out_channels']
    elif name == 'ultralytics':
        fpn, channels = _resolve(('fpns', 'UltralyticsFPN'))(cfg), cfg['model.backbone.out_channels']
    elif name == 'bifpn':
        fpn = _resolve(('fpns', 'get_bifpn'))(cfg)
        channels = [cfg['model.bifpn.out_ch']] * len(cfg['model.backbone.out_channels'])
    else:
        raise Exception('Unknown FPN name')
    cfg['model.fpn.out_channels'], cfg['model.fpn.out_strides'] = channels, strides
    return fpn


def get_rpn(cfg: dict):
    '''
    Detection head for cfg['model.rpn.name'] (reference: models/registry.py:100-116)
    '''
    entry = _HEADS.get(cfg['model.rpn.name'])
    if entry is None:
        raise NotImplementedError()
    return _resolve(entry)(cfg)


def get_det_layer(cfg: dict):
    '''
    The decode layer CLASS for cfg['model.pred_layer']; the caller instantiates it per level
    (reference: models/registry.py:119-146, models/general.py:35-38)
    '''
    entry = _DECODE_LAYERS.get(cfg['model.pred_layer'])
    if entry is None:
        raise NotImplementedError()
    return _resolve(entry)
