"""Image preparation used by api.detection (reference: utils/image_ops.py:11-188).

The PIL forms below are the host path (and the geometry: target sizes, paddings, pad_info); `resample_tables` builds
the coefficient tables with which the device kernel (ops.resize_bilinear_u8) reproduces PIL's bilinear resize -- what
torchvision's tvf.resize does to a PIL image -- bit for bit.  PIL + numpy only: torchvision is not a dependency here.
"""
import numpy as np
import PIL.Image
import torch


def imread_pil(img_path):
    img = PIL.Image.open(img_path)
    if img.mode == 'L':
        img = PIL.Image.fromarray(np.repeat(np.expand_dims(np.array(img), 2), 3, axis=2))
    return img


def _resize(img, size_hw):
    return img.resize((int(size_hw[1]), int(size_hw[0])), PIL.Image.BILINEAR)


def _pad(img, left, top, right, bottom, fill=0):
    out = PIL.Image.new(img.mode, (img.width + left + right, img.height + top + bottom), fill)
    out.paste(img, (left, top))
    return out


def resize_pil(img, img_size, shorter=True):
    '''Resize such that the longer side == img_size (shorter=False), reference :22-35.'''
    if shorter:
        w, h = img.width, img.height          # torchvision int-size semantics: shorter side -> img_size
        if w <= h:
            return _resize(img, (int(img_size * h / w), img_size))
        return _resize(img, (img_size, int(img_size * w / h)))
    imh, imw = img.height, img.width
    factor = img_size / max(imh, imw)
    return _resize(img, (round(imh * factor), round(imw * factor)))


def pad_to_divisible(img, denom):
    '''Zero-pad right/bottom so both sides are divisible by `denom` (reference :38-52).'''
    img_h, img_w = img.height, img.width
    pad_bottom = int(np.ceil(img_h / denom) * denom) - img_h
    pad_right = int(np.ceil(img_w / denom) * denom) - img_w
    assert 0 <= pad_bottom < denom and 0 <= pad_right < denom
    return _pad(img, 0, 0, pad_right, pad_bottom, 0)


def rect_to_square(image, labels, target_size, aug=False):
    '''Resize longer side to target_size and zero-pad to a centred square (reference :55-137,
    the aug=False branch; labels must be None on the inference path).'''
    assert isinstance(image, PIL.Image.Image) and image.mode == 'RGB'
    if aug or labels is not None:
        raise NotImplementedError('augmentation / label transforms are training-side')
    ori_h, ori_w = image.height, image.width
    resize_scale = target_size / max(ori_w, ori_h)
    resized_w, resized_h = int(ori_w * resize_scale), int(ori_h * resize_scale)
    image = _resize(image, (resized_h, resized_w))
    left = (target_size - resized_w) // 2
    top = (target_size - resized_h) // 2
    right = target_size - resized_w - left
    bottom = target_size - resized_h - top
    image = _pad(image, left, top, right, bottom, 0)
    return image, labels, (ori_w, ori_h, left, top, resized_w, resized_h)


def to_tensor(pil_img):
    '''PIL RGB -> float32 [3,H,W] in 0..1 (what tvf.to_tensor returns, api/detection.py:160).'''
    arr = np.array(pil_img.convert("RGB"), dtype=np.uint8)
    return torch.from_numpy(arr).permute(2, 0, 1).contiguous().float().div(255)


def format_tensor_img(t_img, code):
    '''reference :165-188'''
    assert t_img.dim() == 3 and t_img.shape[0] == 3
    if code == 'RGB_1':
        return t_img
    if code == 'RGB_1_norm':
        mean = torch.tensor([0.485, 0.456, 0.406], dtype=t_img.dtype).view(3, 1, 1)
        std = torch.tensor([0.229, 0.224, 0.225], dtype=t_img.dtype).view(3, 1, 1)
        return (t_img - mean) / std
    raise NotImplementedError()


_PRECISION_BITS = 32 - 8 - 2


def resample_tables(in_size, out_size):
    """Pillow's bilinear coefficient tables for one axis, 8-bit fixed-point form (src/libImaging/Resample.c:
    precompute_coeffs + normalize_coeffs_8bpc): bounds int32 [out,2] = (first tap, tap count), kk int32 [out,ksize].
    All arithmetic in float64 in Pillow's order, so the integers are Pillow's integers."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    ss = 1.0 / filterscale
    center = (np.arange(out_size, dtype=np.float64) + 0.5) * scale
    xmin = np.maximum((center - support + 0.5).astype(np.int64), 0)          # C (int) cast: truncation
    xmax = np.minimum((center + support + 0.5).astype(np.int64), in_size) - xmin
    x = np.arange(ksize, dtype=np.float64)[None, :]
    a = np.abs((x + xmin[:, None] - center[:, None] + 0.5) * ss)
    w = np.where(a < 1.0, 1.0 - a, 0.0)
    w[x >= xmax[:, None]] = 0.0
    ww = np.zeros(out_size, np.float64)
    for j in range(ksize):                                                   # Pillow sums the taps in order
        ww = ww + w[:, j]
    w = np.where(ww[:, None] != 0.0, w / np.where(ww == 0.0, 1.0, ww)[:, None], w)
    kk = (0.5 + w * (1 << _PRECISION_BITS)).astype(np.int64).astype(np.int32)     # weights are >= 0 for this filter
    bounds = np.stack([xmin, xmax], axis=1).astype(np.int32)
    return bounds, kk
