"""Box ops named by the hot path (reference: utils/bbox_ops.py:6-49, 309-316)."""
import torch

from .. import ops


def bboxes_iou(bboxes_a, bboxes_b, xyxy=False):
    """Pairwise IoU [N,K] on the GPU, same arithmetic as the reference (chainercv form:
    intersection zeroed unless tl < br on both axes; no clamp, no epsilon)."""
    if bboxes_a.dim() == 1:
        bboxes_a = bboxes_a.unsqueeze(0)
    assert bboxes_a.dim() == bboxes_b.dim() == 2
    if bboxes_a.shape[1] != 4 or bboxes_b.shape[1] != 4:
        raise IndexError()
    if not bboxes_a.is_cuda:
        if not torch.cuda.is_available():
            raise RuntimeError('bboxes_iou runs on MI355X only; no GPU is visible')
        bboxes_a, bboxes_b = bboxes_a.cuda(), bboxes_b.cuda()
    return ops.bboxes_iou(bboxes_a, bboxes_b.to(bboxes_a.device), xyxy=xyxy)


def cxcywh_to_x1y1x2y2(cxcywh):
    """Centre-format boxes [..., >= 4] -> corner format, a new tensor (reference: utils/bbox_ops.py:309-316).  One
    launch of the to-corners kernel of csrc/boxops.hip: float32 `c - s / 2`, `c + s / 2` in the reference's operation
    order, so the result is bit-identical; extra columns (a rotated box's angle) are carried over."""
    assert cxcywh.shape[-1] >= 4
    # the reference returns a tensor of the input's device and dtype for any dtype: the kernel computes in float32
    # on the GPU; the result goes back to where and what the input was (float64 / integer inputs through a float32 round trip)
    src_device, src_dtype = cxcywh.device, cxcywh.dtype
    if not cxcywh.is_cuda:
        if not torch.cuda.is_available():
            raise RuntimeError('cxcywh_to_x1y1x2y2 runs on MI355X only; no GPU is visible')
        cxcywh = cxcywh.cuda()
    out = ops.cxcywh_to_x1y1x2y2(cxcywh if src_dtype == torch.float32 else cxcywh.float())
    return out.to(device=src_device, dtype=src_dtype)
