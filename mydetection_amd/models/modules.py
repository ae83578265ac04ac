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


def prepare_conv(owner, slot, conv, bn, depthwise=False):
    """Kernel-ready parameters of `conv` (+ optional eval-mode `bn` folded into per-channel scale/shift):
    (weights OHWI -- or [k,k,C] for a depthwise conv --, scale or None, shift or None).  Cached on `owner`
    under `slot` and rebuilt when any source tensor is replaced or modified in place."""
    tensors = [conv.weight] + ([bn.weight, bn.bias, bn.running_mean, bn.running_var] if bn is not None else [])
    if conv.bias is not None:
        tensors.append(conv.bias)
    key = _versions(*tensors)
    cache = owner.__dict__.setdefault('_prep_cache', {})
    hit = cache.get(slot)
    if hit is None or hit[0] != key:
        with torch.no_grad():
            w = conv.weight.detach().float()
            if depthwise:
                w = w.permute(2, 3, 0, 1).reshape(w.shape[2], w.shape[3], w.shape[0]).contiguous()   # [C,1,k,k] -> [k,k,C]
            else:
                w = w.permute(0, 2, 3, 1).contiguous()                                               # OIHW -> OHWI
            if bn is not None:
                inv = (bn.running_var.float() + bn.eps).sqrt().reciprocal()
                scale = (bn.weight.float() * inv).contiguous()
                shift = (bn.bias.float() - bn.running_mean.float() * scale).contiguous()
                if conv.bias is not None:
                    shift = (shift + conv.bias.float() * scale).contiguous()
            else:
                scale = None
                shift = conv.bias.detach().float().contiguous() if conv.bias is not None else None
        hit = (key, (w, scale, shift))
        cache[slot] = hit
    return hit[1]


def prepare_wino(owner, slot, w_ohwi):
    """Transform-domain copies of a prepared 3x3 OHWI weight: (ops.wino_weights, ops.wino4_weights), each None when
    its fused Winograd kernel does not cover the shape (F(4x4) only from ops.WINO4_MIN_CIN input channels up).
    Cached on `owner` next to the weight they were made from."""
    cache = owner.__dict__.setdefault('_prep_cache', {})
    hit = cache.get(slot)
    if hit is None or hit[0] is not w_ohwi:
        u = ops.wino_weights(w_ohwi) if ops.WINOGRAD else None
        u4 = ops.wino4_weights(w_ohwi) if ops.WINOGRAD and ops.WINOGRAD4 and w_ohwi.shape[3] >= ops.WINO4_MIN_CIN else None
        hit = (w_ohwi, (u, u4))
        cache[slot] = hit
    return hit[1]


def prepare_b3(owner, slot, w_ohwi):
    """Three-plane bfloat16 copy of a prepared OHWI weight for the split-bf16 implicit GEMM (ops.split_bf16; cached on `owner`
    next to the weight it was made from), or None when no shape of the layer can take that kernel (Cin % 16; 1x1 layers: Cin % 4)."""
    if not ops.SPLIT_BF16 or (w_ohwi.shape[3] % 16 and not (w_ohwi.shape[1] == 1 and w_ohwi.shape[2] == 1 and w_ohwi.shape[3] % 4 == 0)):
        return None
    cache = owner.__dict__.setdefault('_prep_cache', {})
    hit = cache.get(slot)
    if hit is None or hit[0] is not w_ohwi:
        hit = (w_ohwi, ops.split_bf16(w_ohwi))
        cache[slot] = hit
    return hit[1]


class FusedConvMixin:
    """Caches kernel-ready parameters (OHWI weights, per-channel scale/shift)."""

    def _prepared(self, conv, bn):
        return prepare_conv(self, 'main', conv, bn)


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

    def forward(self, x, residual=None, upcat_lo=None):
        """upcat_lo: a half-resolution map; the layer then computes self(cat((nearest_2x(upcat_lo), x), 1)) (reference:
        models/fpns.py:62-66) -- for the 1x1 shapes the fused launch covers without ever writing the concatenated tensor
        (ops.conv1x1_upcat), otherwise through ops.upsample_concat.  Passed through __call__, so module hooks fire on
        either path."""
        if self.training:
            raise NotImplementedError('mydetection_amd implements the inference path only; call model.eval()')
        w, scale, shift = self._prepared(self.conv, self.bn)
        p = (self.k - 1) // 2
        if upcat_lo is not None:
            assert residual is None
            if self.k == 1 and self.s == 1:
                y = ops.conv1x1_upcat(upcat_lo, x, w, scale, shift, ops.ACT_LEAKY)
                if y is not None:
                    return y
            x = ops.upsample_concat(upcat_lo, tuple(x.shape[2:4]), x)        # cat((up(lo), x), dim=1), then the plain layer
        if w.shape[3] == 3 and self.k == 3 and w.shape[0] == 32 and residual is None:
            return ops.conv2d_stem(x, w, scale, shift, self.s, (p, p, p, p), ops.ACT_LEAKY)
        u, u4 = prepare_wino(self, 'wino', w) if self.k == 3 and self.s == 1 else (None, None)
        # the layers that stay on the direct implicit GEMM (1x1, stride-2 3x3) take its split-bf16 form where ops.b3_takes says so
        # ... and the 3x3 layers their patch-resident form where ops.p3_takes says so (stride 2; stride 1 below F(4x4)'s channel limit)
        b3 = prepare_b3(self, 'b3', w) if (u is None and u4 is None) or (self.k == 3 and w.shape[3] <= ops.P3_S1_MAX_CIN) else None
        return ops.conv2d(x, w, scale, shift, self.k, self.s, (p, p, p, p), ops.ACT_LEAKY, residual=residual, wino=u, wino4=u4, b3=b3)


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


class Swish(nn.Module):
    """x * sigmoid(x) (reference: models/modules.py:41-43).  Parameter-free placeholder: every swish on the
    hot path is fused into the producing kernel's epilogue."""
    def forward(self, x):
        raise NotImplementedError('Swish is fused into the producing conv kernel on the inference path')


class SeparableConv2d(nn.Module):
    '''
    Depthwise (no bias) -> pointwise 1x1 (bias), two HIP launches (reference: models/modules.py:5-21).
    `bn`/`act` let the caller fold a following BatchNorm2d / swish into the pointwise epilogue.
    '''
    def __init__(self, in_ch, out_ch, kernel_size, stride, padding):
        super().__init__()
        self.k, self.s, self.p = kernel_size, stride, padding
        self.depthwise = nn.Conv2d(in_ch, in_ch, kernel_size, stride, padding=padding, groups=in_ch, bias=False)
        self.pointwise = nn.Conv2d(in_ch, out_ch, 1, 1, padding=0)

    def fusable(self):
        """True when the fused node kernel (ops.sepconv_nodes) covers this layer: 3x3, stride 1, pad 1, an
        instantiated channel count."""
        return (ops.FUSED_NODES and self.k == 3 and self.s == 1 and self.p == 1
                and self.depthwise.in_channels in ops.SEPCONV_CHANNELS and self.pointwise.out_channels % 4 == 0)

    def node(self, inputs, modes=None, fuse_weights=None, bn=None, act=ops.ACT_NONE, out=None):
        """Descriptor of this layer for `ops.sepconv_nodes` (several layers share one launch)."""
        if self.training:
            raise NotImplementedError('mydetection_amd implements the inference path only; call model.eval()')
        wd, _, _ = prepare_conv(self, 'dw', self.depthwise, None, depthwise=True)
        wp, scale, shift = prepare_conv(self, ('pw', id(bn)), self.pointwise, bn)
        cache = self.__dict__.setdefault('_prep_cache', {})
        hit = cache.get(('pwk', id(bn)))
        if hit is None or hit[0] is not wp:
            hit = (wp, ops.pack_pointwise(wp))
            cache[('pwk', id(bn))] = hit
        return dict(inputs=list(inputs), modes=modes, fuse_weights=fuse_weights, w_dw=wd, w_pw=hit[1], scale=scale,
                    shift=shift, cout=self.pointwise.out_channels, act=act, out=out)

    def node_per_anchor(self, inputs, A, n_cls):
        """Descriptor of this layer as the class tower's last layer for `ops.sepconv_decode_retina`: the pointwise
        weights / bias with every anchor's n_cls rows padded to whole 16-channel blocks (cached)."""
        nd = self.node(inputs)
        wp, _, shift = prepare_conv(self, ('pw', id(None)), self.pointwise, None)
        cache = self.__dict__.setdefault('_prep_cache', {})
        hit = cache.get(('pwk_anchor', A, n_cls))
        if hit is None or hit[0] is not wp:
            hit = (wp, ops.pack_pointwise_per_anchor(wp, shift, A, n_cls))
            cache[('pwk_anchor', A, n_cls)] = hit
        nd['w_pw'], nd['shift'] = hit[1]
        return nd

    def forward(self, x, bn=None, act=ops.ACT_NONE):
        if self.training:
            raise NotImplementedError('mydetection_amd implements the inference path only; call model.eval()')
        if self.fusable():
            return ops.sepconv_nodes([self.node([x], bn=bn, act=act)])[0]
        wd, _, _ = prepare_conv(self, 'dw', self.depthwise, None, depthwise=True)
        x = ops.dwconv(x, wd, None, None, self.k, self.s, (self.p,) * 4, ops.ACT_NONE)
        wp, scale, shift = prepare_conv(self, ('pw', id(bn)), self.pointwise, bn)
        return ops.conv2d(x, wp, scale, shift, 1, 1, (0, 0, 0, 0), act)


class ConvBn(nn.Sequential):
    """nn.Sequential(Conv2d(+bias), BatchNorm2d) evaluated as one fused conv launch
    (reference: conv1x1_bn models/fpns.py:446-450; c5_to_c6 / c6_to_c7 convs models/backbones.py:183-200)."""
    def __init__(self, in_ch, out_ch, k=1, padding=0, eps=0.001):
        super().__init__(nn.Conv2d(in_ch, out_ch, k, stride=1, padding=padding),
                         nn.BatchNorm2d(out_ch, eps=eps, momentum=0.01))
        self.k, self.p = k, padding

    def forward(self, x):
        w, scale, shift = prepare_conv(self, 'main', self[0], self[1])
        # the 3x3 C6 / C7 convs (320->88 @20^2, 88->88 @10^2) take the Winograd kernels like every other 3x3 stride-1 layer
        u, u4 = prepare_wino(self, 'wino', w) if self.k == 3 and self.p == 1 else (None, None)
        return ops.conv2d(x, w, scale, shift, self.k, 1, (self.p,) * 4, ops.ACT_NONE, wino=u, wino4=u4)


class SpconvBn(nn.Sequential):
    """nn.Sequential(SeparableConv2d, BatchNorm2d[, Swish]) with BN (+ swish) folded into the pointwise conv
    (reference: LinearFusion.spconv_bn models/fpns.py:426-429; spconv3x3_bn_swish models/rpns.py:199-205)."""
    def __init__(self, ch, swish):
        mods = [SeparableConv2d(ch, ch, 3, 1, padding=1), nn.BatchNorm2d(ch, eps=0.001, momentum=0.01)]
        if swish:
            mods.append(Swish())
        super().__init__(*mods)
        self.act = ops.ACT_SWISH if swish else ops.ACT_NONE

    def fusable(self):
        return self[0].fusable()

    def node(self, inputs, modes=None, fuse_weights=None, out=None):
        return self[0].node(inputs, modes, fuse_weights, bn=self[1], act=self.act, out=out)

    def forward(self, x):
        return self[0](x, bn=self[1], act=self.act)
