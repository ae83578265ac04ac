"""Oracle (test infrastructure only): conf filter -> top-512 -> class-aware NMS, and box ops.

Restates
  ImageObjects.post_process        utils/structures.py:92-106
  ImageObjects.non_max_suppression utils/structures.py:111-173
  torchvision.ops.nms (CPU kernel) -- third party, PARITY UNPINNED, see oracle/__init__.py
  bboxes_iou                       utils/bbox_ops.py:6-49
  cxcywh_to_x1y1x2y2               utils/bbox_ops.py:309-316
  ImageObjects.bboxes_to_original_ utils/structures.py:175-189
in numpy (float32 arithmetic kept explicit) with the greedy NMS loop in C
(oracle/nms_ref.c) and a pure-Python twin for small cases.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, '_build', 'libnms_ref.so')
        if not os.path.exists(path):
            subprocess.check_call(['make', '-C', _HERE], stdout=subprocess.DEVNULL)
        lib = ctypes.CDLL(path)
        lib.nms_ref_f32.restype = ctypes.c_int
        lib.nms_ref_f32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                    ctypes.c_double, ctypes.c_void_p]
        _LIB = lib
    return _LIB


def nms_single_class(boxes_xyxy, scores, thr):
    """torchvision.ops.nms restatement (C). Returns kept indices (int64) in score order."""
    boxes = np.ascontiguousarray(boxes_xyxy, dtype=np.float32)
    scores = np.ascontiguousarray(scores, dtype=np.float32)
    n = boxes.shape[0]
    keep = np.empty(n, dtype=np.int64)
    if n == 0:
        return keep
    k = _lib().nms_ref_f32(boxes.ctypes.data, scores.ctypes.data, n, float(thr), keep.ctypes.data)
    return keep[:k].copy()


def nms_single_class_py(boxes_xyxy, scores, thr):
    """Pure-Python twin of nms_ref.c (small cases only)."""
    b = np.asarray(boxes_xyxy, dtype=np.float32)
    s = np.asarray(scores, dtype=np.float32)
    n = len(s)
    order = np.argsort(-s, kind='stable')
    areas = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    sup = np.zeros(n, dtype=bool)
    keep = []
    f32 = np.float32
    for _i in range(n):
        i = order[_i]
        if sup[i]:
            continue
        keep.append(i)
        for _j in range(_i + 1, n):
            j = order[_j]
            if sup[j]:
                continue
            w = max(f32(0), min(b[i, 2], b[j, 2]) - max(b[i, 0], b[j, 0]))
            h = max(f32(0), min(b[i, 3], b[j, 3]) - max(b[i, 1], b[j, 1]))
            inter = f32(w * h)
            with np.errstate(invalid='ignore', divide='ignore'):
                ovr = f32(inter / f32(f32(areas[i] + areas[j]) - inter))
            if float(ovr) > float(thr):
                sup[j] = True
    return np.asarray(keep, dtype=np.int64)


def cxcywh_to_x1y1x2y2(b):
    b = np.asarray(b, dtype=np.float32)
    out = b.copy()
    half_w = b[..., 2] / np.float32(2)
    half_h = b[..., 3] / np.float32(2)
    out[..., 0] = b[..., 0] - half_w
    out[..., 1] = b[..., 1] - half_h
    out[..., 2] = b[..., 0] + half_w
    out[..., 3] = b[..., 1] + half_h
    return out


def class_aware_nms(bboxes, cats, scores, nms_thres, bb_format='cxcywh'):
    """Returns positions (into the given arrays) of survivors: class ascending, score descending."""
    if len(scores) == 0:
        return np.zeros(0, dtype=np.int64)
    xyxy = cxcywh_to_x1y1x2y2(bboxes) if bb_format == 'cxcywh' else np.asarray(bboxes, np.float32)
    out = []
    for c in np.unique(cats):                      # sorted ascending, as torch.unique
        pos = np.nonzero(cats == c)[0]
        keep = nms_single_class(xyxy[pos], scores[pos], nms_thres)
        out.append(pos[keep])
    return np.concatenate(out).astype(np.int64)


def post_process(bboxes, cats, scores, conf_thres, nms_thres, topk=512, bb_format='cxcywh'):
    """One image. Returns (bboxes[K,4], cats[K], scores[K], src_idx[K]) where src_idx indexes
    the N input candidates.  Tie order inside torch.topk is unspecified in the reference;
    this oracle breaks score ties by lowest candidate index."""
    bboxes = np.asarray(bboxes, dtype=np.float32)
    cats = np.asarray(cats, dtype=np.int64)
    scores = np.asarray(scores, dtype=np.float32)
    sel = np.nonzero(scores >= np.float32(conf_thres))[0]
    if len(sel) > topk:
        order = np.argsort(-scores[sel], kind='stable')[:topk]
        sel = sel[order]
    keep = class_aware_nms(bboxes[sel], cats[sel], scores[sel], nms_thres, bb_format)
    src = sel[keep]
    return bboxes[src], cats[src], scores[src], src.astype(np.int64)


def bboxes_iou(a, b, xyxy=False):
    a = np.asarray(a, dtype=np.float32)
    b = np.asarray(b, dtype=np.float32)
    if a.ndim == 1:
        a = a[None]
    if a.shape[1] != 4 or b.shape[1] != 4:
        raise IndexError()
    if xyxy:
        tl = np.maximum(a[:, None, :2], b[None, :, :2])
        br = np.minimum(a[:, None, 2:], b[None, :, 2:])
        area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
        area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    else:
        two = np.float32(2)
        tl = np.maximum(a[:, None, :2] - a[:, None, 2:] / two, b[None, :, :2] - b[None, :, 2:] / two)
        br = np.minimum(a[:, None, :2] + a[:, None, 2:] / two, b[None, :, :2] + b[None, :, 2:] / two)
        area_a = a[:, 2] * a[:, 3]
        area_b = b[:, 2] * b[:, 3]
    en = (tl < br).astype(np.float32).prod(axis=2)
    d = br - tl
    area_i = (d[..., 0] * d[..., 1]) * en
    with np.errstate(invalid='ignore', divide='ignore'):
        return area_i / (area_a[:, None] + area_b[None, :] - area_i)


def bboxes_to_original(bboxes, pad_info):
    ori_w, ori_h, tl_x, tl_y, imw, imh = pad_info
    b = np.array(bboxes, dtype=np.float32, copy=True)
    f = np.float32
    b[:, 0] = (b[:, 0] - f(tl_x)) / f(imw) * f(ori_w)
    b[:, 1] = (b[:, 1] - f(tl_y)) / f(imh) * f(ori_h)
    b[:, 2] = b[:, 2] / f(imw) * f(ori_w)
    b[:, 3] = b[:, 3] / f(imh) * f(ori_h)
    return b


def decision_margins(scores, cats, conf, eps=2e-5, topk=512):
    """None when post-processing these candidates cannot hinge on float32 round-off of size eps, else the reason:
    a top-k candidate on the confidence threshold, a tie at the top-k boundary, or two same-class candidates that
    enter NMS with (nearly) equal scores (round-off would decide which suppresses the other).  Used by the fixture
    generator and by the tests that compare detection sets computed from two float32 forwards."""
    order = np.argsort(-scores, kind='stable')
    top = scores[order[:topk + 1]]
    if len(top) and np.abs(top - conf).min() <= eps:
        return 'a score sits on the confidence threshold'
    sel = order[scores[order] >= conf]
    if len(sel) > topk:
        if scores[sel[topk - 1]] - scores[sel[topk]] <= eps:
            return f'tie at the top-{topk} boundary'
        sel = sel[:topk]
    for c in np.unique(cats[sel]):
        s = np.sort(scores[sel][cats[sel] == c])
        if len(s) >= 2 and np.diff(s).min() <= eps:
            return f'two class-{c} candidates with (nearly) equal scores enter NMS'
    return None
