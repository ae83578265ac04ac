"""Oracle (test infrastructure only): numpy restatement of Pillow's bilinear resize of 8-bit images.

The reference resizes with torchvision.transforms.functional.resize on a PIL image (utils/image_ops.py:22-35,
:55-137 rect_to_square; api/detection.py:177-205), which for PIL inputs is PIL.Image.resize(size, BILINEAR).  Pillow's
algorithm (src/libImaging/Resample.c, pinned by the Pillow installed in this image; checked against it in
tests/test_host_cpu.py) is a separable two-pass filter:
  * per output coordinate: scale = in/out, support = 1.0 * max(scale, 1), centre = (xx + 0.5) * scale, taps
    xmin = int(centre - support + 0.5) (clipped at 0) .. xmax = int(centre + support + 0.5) (clipped at in), weights
    triangle((x + xmin - centre + 0.5) / max(scale, 1)) normalised to sum 1 in double precision;
  * 8-bit path: weights -> int (round half away from zero of w * 2^22); out = clip8((2^21 + sum in*w) >> 22);
  * horizontal pass first (to a uint8 image), then the vertical pass; a pass whose size does not change is skipped.
"""
import numpy as np

PRECISION_BITS = 32 - 8 - 2


def coefficients(in_size, out_size):
    """(bounds int32 [out,2] = (xmin, count), kk int32 [out, ksize])."""
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 1.0 * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = np.zeros(ksize, np.float64)
        ww = 0.0
        for x in range(xmax):
            a = abs((x + xmin - center + 0.5) * ss)
            w[x] = 1.0 - a if a < 1.0 else 0.0
            ww += w[x]
        if ww != 0.0:
            w[:xmax] /= ww
        for x in range(ksize):
            v = w[x] * (1 << PRECISION_BITS)
            kk[xx, x] = int(-0.5 + v) if w[x] < 0 else int(0.5 + v)
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _pass(img, out_size, axis):
    in_size = img.shape[axis]
    if in_size == out_size:
        return img
    bounds, kk = coefficients(in_size, out_size)
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((out_size,) + src.shape[1:], np.uint8)
    for xx in range(out_size):
        xmin, n = bounds[xx]
        acc = np.full(src.shape[1:], 1 << (PRECISION_BITS - 1), np.int64)
        for x in range(n):
            acc += src[xmin + x] * int(kk[xx, x])
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def resize_bilinear_u8(img, out_hw):
    """img uint8 [H,W,C] -> uint8 [out_h,out_w,C], as PIL.Image.resize((out_w, out_h), BILINEAR)."""
    tmp = _pass(img, int(out_hw[1]), 1)       # horizontal first
    return _pass(tmp, int(out_hw[0]), 0)
