"""YOLOv3 decode layer (reference: models/detlayers/yolov3.py)."""
import torch
import torch.nn as nn

from ... import ops
from ._common import alloc_outputs, pack_pixel_major


class YOLOLayer(nn.Module):
    '''
    Inference branch of the reference YOLOLayer (models/detlayers/yolov3.py:30-69) as one
    fused HIP kernel: cx=(sigmoid(tx)+x)*stride, cy likewise, w=exp(tw)*anchor_w, h likewise,
    score=sigmoid(conf)*max_c sigmoid(cls_c), class_idx=first argmax; flatten order (a,y,x).
    Unlike the reference the outputs stay in HBM (no .cpu()): post-processing runs there too.
    Training (labels is not None) is out of scope.
    '''
    def __init__(self, level_i: int, cfg: dict):
        super().__init__()
        anchors_all = torch.Tensor(cfg['model.yolo.anchors'])
        indices = torch.Tensor(cfg['model.yolo.anchor_indices'][level_i]).long()
        self.indices = indices
        self.anchors = anchors_all[indices, :]
        self.anch_00wh_all = torch.zeros(len(anchors_all), 4)
        self.anch_00wh_all[:, 2:4] = anchors_all
        self.ignore_thre = cfg['model.yolo.anchor.negative_threshold']
        self.num_anchors = len(indices)
        self.stride = cfg['model.fpn.out_strides'][level_i]
        self.n_cls = cfg['general.num_class']

    def forward(self, raw: dict, img_size, labels=None, _out=None):
        assert isinstance(raw, dict)
        if labels is not None:
            raise NotImplementedError('training/target assignment is outside the inference hot path')
        t_xywh = raw['bbox']
        nB, nA = t_xywh.shape[0], self.num_anchors
        nH, nW = t_xywh.shape[2:4]
        assert t_xywh.shape[1] == nA and t_xywh.shape[-1] == 4
        assert self.n_cls > 0
        packed = getattr(raw, 'packed', None)
        if packed is not None:          # the head's own pixel-major tensors (YOLOHead: one; EfDetHead: box + class)
            box, ldb, bas, bc0 = packed['box']
            cls, ldc, cas, cc0, conf0 = packed['cls']
        else:
            box, ldb, bas = pack_pixel_major([raw['bbox'], raw['conf'], raw['class']], nA)
            cls, ldc, cas, bc0, cc0, conf0 = box, ldb, bas, 0, 5, 4
        n = nA * nH * nW
        if _out is None:
            bbox, cls_idx, score = alloc_outputs(nB, n, box.device)
            n_off = 0
        else:
            bbox, cls_idx, score, n_off = _out
        ops.decode(ops.DECODE_YOLO, box, ldb, bas, bc0, cls, ldc, cas, cc0, conf0, self.anchors.numpy(), nA, self.n_cls,
                   nB, nH, nW, self.stride, tuple(img_size), bbox, cls_idx, score, n_off)
        preds = {
            'bbox': bbox[:, n_off:n_off + n],
            'class_idx': cls_idx[:, n_off:n_off + n],
            'score': score[:, n_off:n_off + n],
        }
        return preds, None

    def _describe(self, raw, img_size):
        """Level descriptor for the single-launch decode (ops.decode_levels), or None."""
        packed = getattr(raw, 'packed', None)
        if packed is None:
            return None
        box, ldb, bas, bc0 = packed['box']
        cls, ldc, cas, cc0, conf0 = packed['cls']
        nH, nW = raw['bbox'].shape[2:4]
        return {'mode': ops.DECODE_YOLO, 'layout': (bas, bc0, cas, cc0, conf0), 'A': self.num_anchors, 'C': self.n_cls,
                'level': {'box': box, 'ldbox': ldb, 'cls': cls, 'ldcls': ldc, 'anchors_wh': self.anchors.numpy(),
                          'H': nH, 'W': nW, 'stride': self.stride}}
