"""Host-side image preparation used by api.detection (reference: utils/image_ops.py:11-188).

Out of scope for kernels this round (SURVEY.md section 8f ranks a fused device-side version
next); implemented with PIL + numpy only because torchvision is not a dependency here.
Resampling follows torchvision's default for PIL inputs (bilinear).
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
