"""Oracle (test infrastructure only): RetinaNet and FCOS2/ATSS inference decode, torch-CPU.

Operate on the reference's raw-dict layout so they can be compared 1:1 with the
imported det layers (tests/golden/detlayers.npz):
  RetinaLayer.forward      models/detlayers/retinanet.py:16-41,56-82   (labels=None)
  FCOS_ATSS_Layer.forward  models/detlayers/fcos2.py:193-251, _ltrb_to :427-458, _xyxy_to_xywh :417-424
  YOLOLayer.forward        models/detlayers/yolov3.py:30-69            (raw-dict form of oracle.yolov3.yolo_decode)
"""
import torch


def retina_anchors(stride, base=4, scales=(1, 1.26, 1.5874), ratios=((1, 1), (1.4, 0.7), (0.7, 1.4))):
    """wh per anchor, scale-major / ratio-minor (retinanet.py:21-28)."""
    base_size = base * stride
    return torch.Tensor([(base_size * sc * rt[0], base_size * sc * rt[1]) for sc in scales for rt in ratios])


def retina_decode(raw, img_size, stride, anchor_wh):
    t = raw['bbox']
    cls_logits = raw['class']
    img_h, img_w = img_size
    nB, nA, nH, nW = t.shape[:4]
    a_cx = torch.arange(stride / 2, img_w, stride).view(1, 1, 1, nW)
    a_cy = torch.arange(stride / 2, img_h, stride).view(1, 1, nH, 1)
    a_wh = anchor_wh.view(1, nA, 1, 1, 2)
    p = torch.empty_like(t).contiguous()
    p[..., 0] = a_cx + t[..., 0] * a_wh[..., 0]
    p[..., 1] = a_cy + t[..., 1] * a_wh[..., 1]
    p[..., 2:4] = torch.exp(t[..., 2:4]) * a_wh
    p[..., 0:4].clamp_(min=1, max=max(img_size))
    score, idx = torch.max(torch.sigmoid(cls_logits), dim=-1)
    n = nA * nH * nW
    return p.view(nB, n, 4), idx.reshape(nB, n), score.reshape(nB, n)


def fcos_decode(raw, img_size, stride):
    t = raw['bbox']
    img_h, img_w = img_size
    nB, nH, nW = t.shape[:3]
    ltrb = torch.exp(t) * stride
    ys = (torch.arange(nH, dtype=torch.float32).view(1, nH, 1)) * stride + stride / 2
    xs = (torch.arange(nW, dtype=torch.float32).view(1, 1, nW)) * stride + stride / 2
    x1 = (xs - ltrb[..., 0]).clamp(min=0, max=img_w)
    y1 = (ys - ltrb[..., 1]).clamp(min=0, max=img_h)
    x2 = (xs + ltrb[..., 2]).clamp(min=0, max=img_w)
    y2 = (ys + ltrb[..., 3]).clamp(min=0, max=img_h)
    xywh = torch.stack([(x1 + x2) / 2, (y1 + y2) / 2, x2 - x1, y2 - y1], dim=-1)
    conf = torch.sigmoid(raw['conf'])
    cls_score, idx = torch.max(torch.sigmoid(raw['class']), dim=3, keepdim=True)
    score = torch.sqrt(conf * cls_score)
    n = nH * nW
    return xywh.view(nB, n, 4), idx.reshape(nB, n), score.reshape(nB, n)


def yolo_decode_raw(raw, stride, anchors_wh):
    t = raw['bbox']
    nB, nA, nH, nW = t.shape[:4]
    p = t.clone().contiguous()
    ys = torch.arange(nH, dtype=torch.float32).view(1, 1, nH, 1)
    xs = torch.arange(nW, dtype=torch.float32).view(1, 1, 1, nW)
    p[..., 0] = (torch.sigmoid(p[..., 0]) + xs) * stride
    p[..., 1] = (torch.sigmoid(p[..., 1]) + ys) * stride
    p[..., 2:4] = torch.exp(p[..., 2:4]) * anchors_wh.view(1, nA, 1, 1, 2)
    conf = torch.sigmoid(raw['conf'])
    cls_score, idx = torch.max(torch.sigmoid(raw['class']), dim=-1, keepdim=True)
    n = nA * nH * nW
    return p.view(nB, n, 4), idx.reshape(nB, n), (conf * cls_score).reshape(nB, n)
