"""RetinaNet decode layer (reference: models/detlayers/retinanet.py)."""
import torch
import torch.nn as nn

from ... import ops
from ._common import alloc_outputs, pack_pixel_major


class RetinaLayer(nn.Module):
    '''
    Inference branch of the reference RetinaLayer (models/detlayers/retinanet.py:16-41,56-82) as one fused
    HIP kernel: anchors wh = base*stride*scale*ratio (scale-major, ratio-minor), centres
    stride/2 + i*stride; cx = acx + tx*aw, w = exp(tw)*aw, all four clamped to [1, max(H,W)];
    score = max_c sigmoid(cls_c) (no objectness).  Outputs stay in HBM.  Training is out of scope.
    '''
    def __init__(self, level_i: int, cfg: dict):
        super().__init__()
        stride = cfg['model.fpn.out_strides'][level_i]
        base_size = cfg['model.retina.anchor.base'] * stride
        scales = cfg['model.retina.anchor.scales']
        ratios = cfg['model.retina.anchor.ratios']
        anchors = []
        for sc in scales:
            for rt in ratios:
                anchors.append((base_size * sc * rt[0], base_size * sc * rt[1]))
        self.anchor_wh = torch.Tensor(anchors)
        self.num_anchors = len(anchors)
        self.positive_thres = cfg['model.retina.anchor.positive_threshold']
        self.negative_thres = cfg['model.retina.anchor.negative_threshold']
        self.stride = stride
        self.n_cls = cfg['general.num_class']
        self.pred_bbox_format = cfg['general.pred_bbox_format']
        if self.pred_bbox_format != 'cxcywh':
            raise NotImplementedError('rotated boxes are outside the inference hot path')
        self.n_bbparam = cfg['general.bbox_param']

    def forward(self, raw: dict, img_size, labels=None, _out=None):
        if labels is not None:
            raise NotImplementedError('training/target assignment is outside the inference hot path')
        stride = self.stride
        img_h, img_w = img_size
        nA = self.num_anchors
        nH, nW = int(img_h / stride), int(img_w / stride)
        nCls = self.n_cls
        t_xywh, cls_logits = raw['bbox'], raw['class']
        nB = t_xywh.shape[0]
        assert t_xywh.shape == (nB, nA, nH, nW, self.n_bbparam)
        assert cls_logits.shape == (nB, nA, nH, nW, nCls)
        packed = getattr(raw, 'packed', None)
        if packed is not None:
            box, ldb, bas, bc0 = packed['box']
            cls, ldc, cas, cc0, _ = packed['cls']
        else:
            box, ldb, bas = pack_pixel_major([t_xywh], nA)
            cls, ldc, cas = pack_pixel_major([cls_logits], nA)
            bc0 = cc0 = 0
        n = nA * nH * nW
        if _out is None:
            bbox, cls_idx, score = alloc_outputs(nB, n, box.device)
            n_off = 0
        else:
            bbox, cls_idx, score, n_off = _out
        ops.decode(ops.DECODE_RETINA, box, ldb, bas, bc0, cls, ldc, cas, cc0, 0, self.anchor_wh.numpy(), nA, nCls,
                   nB, nH, nW, stride, (img_h, img_w), bbox, cls_idx, score, n_off)
        preds = {'bbox': bbox[:, n_off:n_off + n], 'class_idx': cls_idx[:, n_off:n_off + n],
                 'score': score[:, n_off:n_off + n]}
        return preds, None

    def _describe(self, raw, img_size):
        """Level descriptor for the single-launch decode (ops.decode_levels), or None."""
        packed = getattr(raw, 'packed', None)
        if packed is None:
            return None
        box, ldb, bas, bc0 = packed['box']
        cls, ldc, cas, cc0, _ = packed['cls']
        nH, nW = raw['bbox'].shape[2:4]
        return {'mode': ops.DECODE_RETINA, 'layout': (bas, bc0, cas, cc0, 0), 'A': self.num_anchors, 'C': self.n_cls,
                'level': {'box': box, 'ldbox': ldb, 'cls': cls, 'ldcls': ldc, 'anchors_wh': self.anchor_wh.numpy(),
                          'H': nH, 'W': nW, 'stride': self.stride}}
