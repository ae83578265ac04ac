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


def build_reference_model(config_name):
    """Reference OneStageBBox for configs/<name>.json filled with the synthetic weights."""
    import json
    import io
    install()
    from mydetection_amd import synth
    cfg = json.load(open(f'{REFERENCE_ROOT}/configs/{config_name}.json'))
    with no_pretrained(), contextlib.redirect_stdout(io.StringIO()):
        from models.general import OneStageBBox
        model = OneStageBBox(cfg)
    model.load_state_dict(synth.make_state_dict(model.state_dict(), config_name), strict=True)
    model.eval()
    return model, cfg
