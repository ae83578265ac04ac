"""FCOS-style anchor-free decode layers (reference: models/detlayers/fcos2.py)."""
import torch.nn as nn

from ... import ops
from ._common import alloc_outputs, pack_pixel_major


class _FCOSInference(nn.Module):
    '''
    The inference branch shared by the reference's three FCOS-style layers (fcos.py:21-68, fcos2.py:24-69,
    fcos2.py:222-251 are the same arithmetic; they differ in training-time assignment only):
    ltrb = exp(t)*stride around the cell centre x*stride + stride/2, corners clamped to the image, converted to
    cxcywh; score = sqrt(sigmoid(centerness) * max_c sigmoid(cls_c)); order (y, x).
    One fused HIP kernel; outputs stay in HBM.  `conf_key` names the centerness entry of the raw dict.
    '''
    conf_key = 'conf'

    def forward(self, raw, img_size, labels=None, _out=None):
        if labels is not None:
            raise NotImplementedError('training/target assignment is outside the inference hot path')
        stride = self.stride
        img_h, img_w = img_size
        nH, nW = int(img_h / stride), int(img_w / stride)
        nCls = self.n_cls
        assert isinstance(raw, dict)
        t_ltrb, conf_logits, cls_logits = raw['bbox'], raw[self.conf_key], raw['class']
        nB = t_ltrb.shape[0]
        assert t_ltrb.shape == (nB, nH, nW, 4)
        assert conf_logits.shape == (nB, nH, nW, 1)
        assert cls_logits.shape == (nB, nH, nW, nCls)
        packed = getattr(raw, 'packed', None)
        if packed is not None:
            box, ldb, bas, bc0 = packed['box']
            cls, ldc, cas, cc0, conf0 = packed['cls']
        else:
            box, ldb, bas = pack_pixel_major([t_ltrb], 1)
            cls, ldc, cas = pack_pixel_major([conf_logits, cls_logits], 1)
            bc0, cc0, conf0 = 0, 1, 0
        n = nH * nW
        if _out is None:
            bbox, cls_idx, score = alloc_outputs(nB, n, box.device)
            n_off = 0
        else:
            bbox, cls_idx, score, n_off = _out
        ops.decode(ops.DECODE_FCOS, box, ldb, bas, bc0, cls, ldc, cas, cc0, conf0, None, 1, nCls, nB, nH, nW, stride,
                   (img_h, img_w), bbox, cls_idx, score, n_off)
        preds = {'bbox': bbox[:, n_off:n_off + n], 'class_idx': cls_idx[:, n_off:n_off + n],
                 'score': score[:, n_off:n_off + n]}
        return preds, None

    def _describe(self, raw, img_size):
        """Level descriptor for the single-launch decode (ops.decode_levels), or None."""
        packed = getattr(raw, 'packed', None)
        if packed is None:
            return None
        box, ldb, bas, bc0 = packed['box']
        cls, ldc, cas, cc0, conf0 = packed['cls']
        nH, nW = raw['bbox'].shape[1:3]
        return {'mode': ops.DECODE_FCOS, 'layout': (bas, bc0, cas, cc0, conf0), 'A': 1, 'C': self.n_cls,
                'level': {'box': box, 'ldbox': ldb, 'cls': cls, 'ldcls': ldc, 'anchors_wh': None,
                          'H': nH, 'W': nW, 'stride': self.stride}}


class FCOSLayer(_FCOSInference):
    '''
    'FCOS2' (reference: models/detlayers/fcos2.py:11-69).  Training (:70-190) is out of scope.
    '''
    def __init__(self, level_i: int, cfg: dict):
        super().__init__()
        self.anch_min = cfg['model.fcos.anchors'][level_i]
        self.anch_max = cfg['model.fcos.anchors'][level_i + 1]
        self.stride = cfg['model.fpn.out_strides'][level_i]
        self.n_cls = cfg['general.num_class']
        self.center_region = 0.5
        self.ltrb_setting = 'exp_sl1'
        self.ignore_thre = cfg['model.fcos2.ignored_threshold']
        self.bb_format = cfg['general.pred_bbox_format']


class FCOS_ATSS_Layer(_FCOSInference):
    '''
    'FCOS2_ATSS' (reference: models/detlayers/fcos2.py:193-251; ATSS only changes training, :253-405, out of scope).
    '''
    def __init__(self, level_i: int, cfg: dict):
        super().__init__()
        self.strides_all = cfg['model.fpn.out_strides']
        self.stride = cfg['model.fpn.out_strides'][level_i]
        self.n_cls = cfg['general.num_class']
        self.anchors_all = cfg['model.atss.anchors']
        self.anchor = self.anchors_all[level_i]
        self.topk = cfg['model.atss.topk_per_level']
        self.ltrb_setting = 'exp_sl1'
        self.ignore_thre = cfg['model.fcos2.ignored_threshold']
