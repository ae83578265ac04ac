"""Network blocks of the hot path, mirroring models/modules.py of the reference.

Same class names, constructor arguments and ``state_dict`` keys; ``forward`` is one
fused HIP launch (conv as implicit GEMM on FP32 MFMA + folded BN + activation
[+ residual]) instead of three ATen calls.  torch.nn.Conv2d / BatchNorm2d objects are
kept purely as parameter containers so checkpoints load unchanged
(api/detection.py:42-43); they are never called.
Inference only: BN uses running statistics (the reference evaluates under
model.eval(), api/detection.py:33).
"""
import torch
import torch.nn as nn

from .. import ops


def _versions(*tensors):
    return tuple((t.data_ptr(), t._version) for t in tensors)


class FusedConvMixin:
    """Caches kernel-ready parameters (OHWI weights, per-channel scale/shift)."""
    _prep = None
    _prep_key = None

    def _prepared(self, conv, bn):
        tensors = [conv.weight] + ([bn.weight, bn.bias, bn.running_mean, bn.running_var] if bn is not None
                                   else ([conv.bias] if conv.bias is not None else []))
        key = _versions(*tensors)
        if self._prep_key != key:
            with torch.no_grad():
                w = conv.weight.detach().permute(0, 2, 3, 1).contiguous().float()     # OIHW -> OHWI
                if bn is not None:
                    inv = (bn.running_var.float() + bn.eps).sqrt().reciprocal()
                    scale = (bn.weight.float() * inv).contiguous()
                    shift = (bn.bias.float() - bn.running_mean.float() * scale).contiguous()
                    if conv.bias is not None:
                        shift = (shift + conv.bias.float() * scale).contiguous()
                else:
                    scale = None
                    shift = conv.bias.detach().float().contiguous() if conv.bias is not None else None
            self._prep = (w, scale, shift)
            self._prep_key = key
        return self._prep


class ConvBnLeaky(nn.Module, FusedConvMixin):
    '''
    Conv2d + BatchNorm + LeakyReLU(0.1) as one kernel  (reference: models/modules.py:76-95)

    Args:
        c1: input channel, c2: output channel, k: kernel size, s: stride
    '''
    def __init__(self, c1, c2, k=1, s=1):
        super().__init__()
        self.k, self.s = k, s
        self.conv = nn.Conv2d(c1, c2, k, s, padding=(k - 1) // 2, bias=False)
        self.bn = nn.BatchNorm2d(c2, eps=1e-5, momentum=0.01)

    def forward(self, x, residual=None):
        if self.training:
            raise NotImplementedError('mydetection_amd implements the inference path only; call model.eval()')
        w, scale, shift = self._prepared(self.conv, self.bn)
        p = (self.k - 1) // 2
        if w.shape[3] == 3 and self.k == 3 and w.shape[0] == 32 and residual is None:
            return ops.conv2d_stem(x, w, scale, shift, self.s, (p, p, p, p), ops.ACT_LEAKY)
        return ops.conv2d(x, w, scale, shift, self.k, self.s, (p, p, p, p), ops.ACT_LEAKY, residual=residual)


class DarkBlock(nn.Module):
    '''
    Residual block in Darknet53: x + cbl_1(cbl_0(x)); the add rides in cbl_1's epilogue
    (reference: models/modules.py:56-73)
    '''
    def __init__(self, in_out, hidden):
        super().__init__()
        self.cbl_0 = ConvBnLeaky(in_out, hidden, k=1, s=1)
        self.cbl_1 = ConvBnLeaky(hidden, in_out, k=3, s=1)

    def forward(self, x):
        return self.cbl_1(self.cbl_0(x), residual=x)
