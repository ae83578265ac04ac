"""Oracle tooling (build container only): make /root/reference importable.

The reference imports third-party packages this image lacks (torchvision, cv2,
pycocotools, fvcore).  They are replaced by empty stub modules
in ``sys.modules`` -- no reference file is copied or edited -- and the two
checkpoint loaders that need files/network (`models/registry.py:15` torch.load of
weights/dark53_imgnet.pth, `external/efficientnet/utils.py:323-335` model_zoo
download) are intercepted.  The only behaviour supplied from outside is
``torchvision.ops.nms`` = oracle.postprocess.nms_single_class (parity unpinned
there, see oracle/__init__.py).

Self-skips (raises ReferenceUnavailable) when /root/reference is absent, e.g. on
the GPU box.
"""
import contextlib
import os
import sys
import types

import numpy as np
import torch

REFERENCE_ROOT = '/root/reference'


class ReferenceUnavailable(RuntimeError):
    pass


def _stub(name):
    m = types.ModuleType(name)
    sys.modules[name] = m
    return m


def install():
    if not os.path.isdir(REFERENCE_ROOT):
        raise ReferenceUnavailable(REFERENCE_ROOT + ' not present')
    if getattr(install, '_done', False):
        return
    from . import postprocess as pp
    tv = _stub('torchvision')
    tv.transforms = _stub('torchvision.transforms')
    tv.transforms.functional = _stub('torchvision.transforms.functional')
    tv.ops = _stub('torchvision.ops')

    def nms(boxes, scores, iou_threshold):
        keep = pp.nms_single_class(boxes.detach().cpu().numpy(), scores.detach().cpu().numpy(),
                                   float(iou_threshold))
        return torch.from_numpy(keep)
    tv.ops.nms = nms
    # torchvision.transforms.functional for PIL inputs is a thin layer over PIL itself (resize -> Image.resize with
    # BILINEAR, pad -> a new image with the old one pasted in, to_tensor -> uint8 HWC / 255 as float32 CHW, normalize ->
    # sub_(mean).div_(std)); torchvision is not installed, so those four are supplied with exactly that behaviour and the reference's own
    # utils/image_ops.py (resize_pil, pad_to_divisible, rect_to_square, format_tensor_img) runs on top of them
    import PIL.Image

    def tv_resize(img, size, interpolation=None):
        if isinstance(size, int):
            w, h = img.size
            if w <= h:
                size = (int(size * h / w), size)
            else:
                size = (size, int(size * w / h))
        return img.resize((int(size[1]), int(size[0])), PIL.Image.BILINEAR)

    def tv_pad(img, padding, fill=0, padding_mode='constant'):
        left, top, right, bottom = padding
        fill = 0 if fill is None else fill              # torchvision's _parse_fill
        out = PIL.Image.new(img.mode, (img.width + left + right, img.height + top + bottom), fill)
        out.paste(img, (left, top))
        return out

    def tv_to_tensor(img):
        arr = np.array(img.convert('RGB'), dtype=np.uint8)
        return torch.from_numpy(arr).permute(2, 0, 1).contiguous().float().div(255)
    def tv_normalize(tensor, mean, std, inplace=False):
        mean = torch.as_tensor(mean, dtype=tensor.dtype).view(-1, 1, 1)
        std = torch.as_tensor(std, dtype=tensor.dtype).view(-1, 1, 1)
        return tensor.clone().sub_(mean).div_(std)
    tv.transforms.functional.normalize = tv_normalize
    tv.transforms.functional.resize = tv_resize
    tv.transforms.functional.pad = tv_pad
    tv.transforms.functional.to_tensor = tv_to_tensor
    _stub('cv2')
    pc = _stub('pycocotools')
    pc.mask = _stub('pycocotools.mask')
    pc.cocoeval = _stub('pycocotools.cocoeval')
    pc.cocoeval.COCOeval = object
    fv = _stub('fvcore')
    fv.nn = _stub('fvcore.nn')
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    install._done = True


@contextlib.contextmanager
def no_pretrained():
    """Neutralise checkpoint loading while a reference model is constructed."""
    orig_load = torch.load
    orig_lsd = torch.nn.Module.load_state_dict
    torch.load = lambda *a, **k: {}
    torch.nn.Module.load_state_dict = lambda self, sd, strict=True: None
    patched = None
    try:
        try:
            import external.efficientnet.model as efm
            patched = (efm, efm.load_pretrained_weights)
            efm.load_pretrained_weights = lambda *a, **k: None
        except Exception:
            patched = None
        yield
    finally:
        torch.load = orig_load
        torch.nn.Module.load_state_dict = orig_lsd
        if patched:
            patched[0].load_pretrained_weights = patched[1]


# configurations composed through the reference's registry that it has no JSON file for: (base file, overrides)
DERIVED = {'d1_fcs2_p3': ('d1_fcs2', {'model.backbone.num_levels': 3, 'model.fcos.anchors': [0, 64, 128, 100000000]})}


def reference_config(config_name):
    import json
    base, over = DERIVED.get(config_name, (config_name, {}))
    cfg = json.load(open(f'{REFERENCE_ROOT}/configs/{base}.json'))
    cfg.update(over)
    return cfg


def build_reference_model(config_name, recipe='conditioned'):
    """Reference OneStageBBox for configs/<name>.json (or a DERIVED composition) filled with the synthetic weights."""
    import io
    install()
    from mydetection_amd import synth
    cfg = reference_config(config_name)
    with no_pretrained(), contextlib.redirect_stdout(io.StringIO()):
        from models.general import OneStageBBox
        model = OneStageBBox(cfg)
    model.load_state_dict(synth.make_state_dict(model.state_dict(), config_name, recipe), strict=True)
    model.eval()
    return model, cfg
