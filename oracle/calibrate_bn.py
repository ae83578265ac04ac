"""Build-container tool: measure BatchNorm running statistics for the EfficientDet-family configs.

    python -m oracle.calibrate_bn efficientdet-d1 d1_fcs2_atss
    python -m oracle.calibrate_bn --stiff efficientdet-d1 d1_fcs2_atss      # the ill-conditioned second parameter set

Runs the imported reference model (eval mode) on two synthetic 512x512 images with a forward
pre-hook on every BatchNorm2d that sets its running_mean / running_var to the batch statistics of
its input before it is applied, so layers are calibrated in execution order in one pass.  Writes
mydetection_amd/calib/<config>.npz (data only).  The statistics are part of the synthetic-weight
recipe (mydetection_amd/synth.py), not of the reference.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import _refimport  # noqa: E402
from mydetection_amd import synth  # noqa: E402

MEAN = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
STD = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)


def calibrate(name, recipe='conditioned'):
    ck = name if recipe == 'conditioned' else f'{name}.{recipe}'
    synth._CALIB_CACHE[ck] = {}                         # start from the uncalibrated recipe
    model, cfg = _refimport.build_reference_model(name, recipe)
    x = synth.make_images(2, 512, seed=100)
    if cfg['general.input_format'] == 'RGB_1_norm':
        x = (x - MEAN) / STD

    def pre_hook(mod, inp):
        t = inp[0]
        mod.running_mean.copy_(t.mean(dim=(0, 2, 3)))
        mod.running_var.copy_(t.var(dim=(0, 2, 3), unbiased=False).clamp_min(1e-4))
    for mod in model.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.register_forward_pre_hook(pre_hook)
    # spread of every final head layer's output under unit gain (synth._efdet_last divides by it)
    spread = {}

    def out_hook(name):
        def f(m, _i, o):
            conv = m.pointwise if hasattr(m, 'pointwise') else m
            o = o - conv.bias.view(1, -1, 1, 1)
            spread['__std__/' + name] = np.float32(o.std(dim=(0, 2, 3), unbiased=False).pow(2).mean().sqrt().item())
            spread['__mean__/' + name] = o.mean(dim=(0, 2, 3)).numpy().astype(np.float32)
        return f
    finals = {k.rsplit('.', 2)[0] if '.pointwise.' in k else k.rsplit('.', 1)[0]
              for k in model.state_dict() if synth._efdet_last_kind(k) and '.depthwise.' not in k}
    for mname, mod in model.named_modules():
        if mname in finals:
            mod.register_forward_hook(out_hook(mname))
    with torch.no_grad():
        model(x)
    assert len(spread) == 2 * len(finals)
    out = {k: v.numpy().astype(np.float32) for k, v in model.state_dict().items()
           if k.endswith(('running_mean', 'running_var'))}
    out.update(spread)
    os.makedirs(os.path.join(ROOT, 'mydetection_amd', 'calib'), exist_ok=True)
    np.savez_compressed(os.path.join(ROOT, 'mydetection_amd', 'calib', ck + '.npz'), **out)
    synth._CALIB_CACHE.pop(ck, None)
    print(ck, len(out), 'tensors', sum(v.size for v in out.values()), 'floats')


if __name__ == '__main__':
    torch.set_num_threads(8)
    args = [a for a in sys.argv[1:] if a != '--stiff']
    for n in args or ['efficientdet-d1', 'd1_fcs2_atss']:
        calibrate(n, 'stiff' if '--stiff' in sys.argv else 'conditioned')
