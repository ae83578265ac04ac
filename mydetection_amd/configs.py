"""Model configurations of the hot path, by name.

The reference keeps one JSON file per model under configs/ and `models.general.name_to_model(name)` loads it
(models/general.py:9-16); every factory of the plug-in surface then reads its own keys from that flat dict and
writes the derived ones ('PLACEHOLDER' entries, models/registry.py:26-39,60-74).  Here the dicts are composed from
shared fragments -- only the keys the inference path reads are present; training keys do not exist in this package.
A file `configs/<name>.json` at the project root, when present, takes precedence (a maintainer's own model).
"""
import copy

_PLACEHOLDER = 'PLACEHOLDER'        # filled in by get_backbone / get_fpn


def _general(input_format, divisibility):
    return {'base': 'OneStageBBox',
            'general.input_format': input_format,          # utils/image_ops.py:165-188
            'general.input_divisibility': divisibility,    # api/detection.py:49
            'general.num_class': 80,
            'general.pred_bbox_format': 'cxcywh',
            'general.bbox_param': 4}


def _pyramid(backbone, levels, fpn):
    return {'model.backbone.name': backbone, 'model.backbone.num_levels': levels,
            'model.backbone.out_channels': _PLACEHOLDER, 'model.backbone.out_strides': _PLACEHOLDER,
            'model.fpn.name': fpn, 'model.fpn.out_channels': _PLACEHOLDER, 'model.fpn.out_strides': _PLACEHOLDER}


def _test(input_size, nms, conf=0.5, **extra):
    return dict({'test.preprocessing': 'resize_pad_divisible', 'test.default_input_size': input_size,
                 'test.default_conf_thres': conf, 'test.ap_conf_thres': 0.005, 'test.nms_thres': nms}, **extra)


def _yolo_anchors(wh, per_level=3):
    n = len(wh) // per_level
    return {'model.yolo.num_anchor_per_level': per_level, 'model.yolo.anchors': [list(a) for a in wh],
            'model.yolo.anchor_indices': [list(range(per_level * i, per_level * (i + 1))) for i in range(n)],
            'model.yolo.anchor.negative_threshold': 0.7}


def _efficientnet_bifpn(divisibility, dropout, c6c7=None, levels=5):
    """EfficientNet-B1 with five levels (C6/C7 appended, 88 channels) under four BiFPN5 layers, or its three backbone
    levels under four BiFPN3 layers."""
    cfg = _general('RGB_1_norm', divisibility)
    cfg.update(_pyramid('efficientnet-b1', levels, 'bifpn'))
    cfg.update({'model.backbone.C6C7_out_channels': 88, 'model.efficientnet.enable_dropout': dropout,
                'model.bifpn.out_ch': 88, 'model.bifpn.repeat_num': 4, 'model.bifpn.fusion_method': 'linear'})
    if c6c7:
        cfg['model.efficientnet.C6C7_downsample'] = c6c7      # models/backbones.py:182
    return cfg


def _effrpn(name, anchors, conf, **extra):
    return dict({'model.rpn.name': name, 'model.effrpn.repeat_num': 3, 'model.effrpn.num_anchor_per_level': anchors,
                 'model.effrpn.enable_conf': conf}, **extra)


_FCOS_RANGES = [0, 64, 128, 256, 512, 100000000]


def _build():
    out = {}

    c = _general('RGB_1', 32)
    c.update(_pyramid('dark53', 3, 'yolov3'))
    c.update({'model.rpn.name': 'yolov3', 'model.pred_layer': 'YOLO'})
    c.update(_yolo_anchors([(10, 13), (16, 30), (33, 23), (30, 61), (62, 45), (59, 119), (116, 90), (156, 198), (373, 326)]))
    c.update(_test(608, 0.45))
    out['yolov3_80'] = c

    c = _efficientnet_bifpn(128, True)
    c.update(_effrpn('effrpn', 9, False))
    c.update({'model.pred_layer': 'RetinaNet', 'model.retina.num_anchor_per_level': 9, 'model.retina.anchor.base': 4,
              'model.retina.anchor.scales': [1, 1.26, 1.5874], 'model.retina.anchor.ratios': [[1, 1], [1.4, 0.7], [0.7, 1.4]],
              'model.retina.anchor.positive_threshold': 0.5, 'model.retina.anchor.negative_threshold': 0.5})
    # input_divisibility / preprocessing are absent from the reference's file (its Detector raises KeyError)
    c.update(_test(640, 0.5, **{'test.to_square': True}))
    out['efficientdet-d1'] = c

    for name, layer, extra in (('d1_fcs2_atss', 'FCOS2_ATSS', {'model.atss.anchors': [24, 48, 96, 192, 384],
                                                             'model.atss.topk_per_level': 9}),
                               ('d1_fcs2', 'FCOS2', {'model.fcos.anchors': _FCOS_RANGES})):
        c = _efficientnet_bifpn(128, False, 'conv')
        c.update(_effrpn('effrpn', 1, True, **{'model.effrpn.cls_last': 'conv'}))
        c.update({'model.pred_layer': layer, 'model.fcos2.ignored_threshold': 0.7}, **extra)
        c.update(_test(640, 0.5))
        out[name] = c

    c = _efficientnet_bifpn(128, False)
    c.update(_effrpn('effrpn_ct', 1, False, **{'model.effrpn.enable_centerscore': True}))
    c.update({'model.pred_layer': 'FCOS', 'model.fcos.anchors': _FCOS_RANGES})
    c.update(_test(640, 0.5))
    out['d1_fcs'] = c

    # registry composition on three pyramid levels (get_bifpn -> BiFPN3, models/fpns.py:302-303; the reference ships it
    # for its rotated-box model d1_rapid): d1_fcs2 with model.backbone.num_levels = 3
    c = _efficientnet_bifpn(32, False, None, levels=3)
    c.update(_effrpn('effrpn', 1, True, **{'model.effrpn.cls_last': 'conv'}))
    c.update({'model.pred_layer': 'FCOS2', 'model.fcos2.ignored_threshold': 0.7, 'model.fcos.anchors': _FCOS_RANGES[:3] + [100000000]})
    c.update(_test(640, 0.5))
    out['d1_fcs2_p3'] = c

    c = _efficientnet_bifpn(32, True)
    c.update(_effrpn('effrpn', 3, True))
    c.update({'model.pred_layer': 'YOLO'})
    c.update(_yolo_anchors([(12.6, 13.2), (23.5, 38.1), (57.3, 32.3), (42.9, 75.5), (106.6, 61.2), (60.4, 123.5),
                            (84.5, 191.6), (131.9, 123.9), (212.4, 85.6), (125.4, 278.9), (179.6, 196.4), (347.0, 107.1),
                            (272.3, 199.2), (238.8, 321.5), (373.1, 258.9)]))
    c.update(_test(640, 0.5))
    out['d1_yv3'] = c

    # YOLOv5-m trunk + pyramid under the YOLOv3 head / decode, and under the anchor-free FCOS2 decode
    # (configs/u5m_yv3.json, configs/u5m_fcs2.json)
    ul = {'model.ultralytics.first': 'Focus', 'model.ultralytics.depth_muliple': 0.67, 'model.ultralytics.channel_muliple': 0.75}
    c = _general('RGB_1', 32)
    c.update(_pyramid('ultralytics', 3, 'ultralytics'))
    c.update(ul)
    c.update({'model.rpn.name': 'yolov3', 'model.pred_layer': 'YOLO'})
    c.update(_yolo_anchors([(10, 13), (16, 30), (33, 23), (30, 61), (62, 45), (59, 119), (116, 90), (156, 198), (373, 326)]))
    c.update(_test(640, 0.45))
    out['u5m_yv3'] = c

    c = _general('RGB_1', 32)
    c.update(_pyramid('ultralytics', 3, 'ultralytics'))
    c.update(ul)
    c.update({'model.rpn.name': 'yolov3', 'model.yolo.num_anchor_per_level': 1, 'model.pred_layer': 'FCOS2',
              'model.fcos.anchors': [0, 64, 128, 100000000], 'model.fcos2.ignored_threshold': 0.7})
    c.update(_test(640, 0.45))
    out['u5m_fcs2'] = c
    return out


_CONFIGS = _build()
NAMES = tuple(_CONFIGS)


def get(model_name):
    """A fresh dict for `model_name` (the factories write into it); FileNotFoundError for an unknown name, which is
    what the reference's `open('configs/<name>.json')` raises."""
    if model_name not in _CONFIGS:
        raise FileNotFoundError(f"no configuration named '{model_name}' (known: {', '.join(NAMES)})")
    return copy.deepcopy(_CONFIGS[model_name])
