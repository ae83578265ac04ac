"""Feature pyramids of the hot path (reference: models/fpns.py)."""
import torch
import torch.nn as nn

from .. import ops
from .modules import ConvBn, ConvBnLeaky, SpconvBn, prepare_conv


class YOLOv3FPN(nn.Module):
    '''
    YOLOv3 top-down pyramid (reference: models/fpns.py:6-25): P5 first, each finer level
    receives the coarser level's cbl_4 output.
    '''
    def __init__(self, cfg: dict):
        super().__init__()
        assert cfg['model.backbone.num_levels'] == 3
        ch3, ch4, ch5 = cfg['model.backbone.out_channels']
        self.branch_P3 = YOLOBranch(ch3, prev_ch=(ch4 // 2, ch3 // 2))
        self.branch_P4 = YOLOBranch(ch4, prev_ch=(ch5 // 2, ch4 // 2))
        self.branch_P5 = YOLOBranch(ch5)

    def forward(self, features):
        c3, c4, c5 = features
        p5, c5_to_c4 = self.branch_P5(c5, previous=None)
        p4, c4_to_c3 = self.branch_P4(c4, previous=c5_to_c4)
        p3, _ = self.branch_P3(c3, previous=c4_to_c3)
        return [p3, p4, p5]


class YOLOBranch(nn.Module):
    '''
    One pyramid level (reference: models/fpns.py:27-74): optional 1x1 'process' of the coarser
    feature -> nearest upsample + concat (one fused kernel) -> 6 ConvBnLeaky.
    Returns (cbl_5 output, cbl_4 output).
    '''
    def __init__(self, in_, prev_ch=None):
        super().__init__()
        assert in_ % 2 == 0, 'input channel must be divisible by 2'
        if prev_ch:
            self.process = ConvBnLeaky(prev_ch[0], prev_ch[1], k=1, s=1)
            in_after_cat = in_ + prev_ch[1]
        else:
            in_after_cat = in_
        self.cbl_0 = ConvBnLeaky(in_after_cat, in_ // 2, k=1, s=1)
        self.cbl_1 = ConvBnLeaky(in_ // 2, in_, k=3, s=1)
        self.cbl_2 = ConvBnLeaky(in_, in_ // 2, k=1, s=1)
        self.cbl_3 = ConvBnLeaky(in_ // 2, in_, k=3, s=1)
        self.cbl_4 = ConvBnLeaky(in_, in_ // 2, k=1, s=1)
        self.cbl_5 = ConvBnLeaky(in_ // 2, in_, k=3, s=1)

    def forward(self, x, previous=None):
        if previous is not None:
            pre = self.process(previous)
            x = self.cbl_0(x, upcat_lo=pre)           # cbl_0(cat((up2x(pre), x), 1)): one launch where the shape allows, else two
        else:
            x = self.cbl_0(x)
        x = self.cbl_1(x)
        x = self.cbl_2(x)
        x = self.cbl_3(x)
        feature = self.cbl_4(x)
        x = self.cbl_5(feature)
        return x, feature


class UltralyticsFPN(nn.Module):
    """YOLOv5 top-down pyramid (reference: models/fpns.py:77-107): P5 = CSP(C5); P4 = CSP(Conv1x1(cat(up2x(P5), C4)));
    P3 likewise from P4.  Upsampling and concatenation are one launch.  The CSP depth is scaled with the CHANNEL
    multiple, as the reference does (models/fpns.py:88)."""
    def __init__(self, global_cfg):
        super().__init__()
        from ..external.ultralytics.common import BottleneckCSP, Conv
        assert global_cfg['model.backbone.num_levels'] == 3
        ch3, ch4, ch5 = global_cfg['model.backbone.out_channels']
        chm = global_cfg['model.ultralytics.channel_muliple']
        n = max(round(3 * chm), 1)
        self.to_p5 = BottleneckCSP(ch5, ch5, n=n, shortcut=False)
        self.to_p4 = nn.Sequential(Conv(ch4 + ch5, ch4, k=1, s=1), BottleneckCSP(ch4, ch4, n=n, shortcut=False))
        self.to_p3 = nn.Sequential(Conv(ch3 + ch4, ch3, k=1, s=1), BottleneckCSP(ch3, ch3, n=n, shortcut=False))

    def forward(self, features):
        c3, c4, c5 = features
        p5 = self.to_p5(c5)
        p4 = self.to_p4(ops.upsample_concat(p5, (p5.shape[2] * 2, p5.shape[3] * 2), c4))      # cat([up(p5), c4], 1)
        p3 = self.to_p3(ops.upsample_concat(p4, (p4.shape[2] * 2, p4.shape[3] * 2), c3))
        return [p3, p4, p5]


def get_bifpn(cfg: dict):
    '''repeat_num stacked BiFPN layers; only the first projects the backbone channels (reference: models/fpns.py:294-312)'''
    in_channels = cfg['model.backbone.out_channels']
    out_ch = cfg['model.bifpn.out_ch']
    repeat_num = cfg['model.bifpn.repeat_num']
    fusion_method = cfg['model.bifpn.fusion_method']
    assert repeat_num >= 1
    if len(in_channels) == 3:
        fpn_func = BiFPN3
    elif len(in_channels) == 5:
        fpn_func = BiFPN5
    else:
        raise NotImplementedError()
    fpn = [fpn_func(out_ch, fusion_method=fusion_method, in_chs=in_channels)]
    for _ in range(repeat_num - 1):
        fpn.append(fpn_func(out_ch, fusion_method=fusion_method))
    return nn.Sequential(*fpn)


def conv1x1_bn(in_ch, out_ch):
    return ConvBn(in_ch, out_ch, 1, 0)


def _identity(x):
    return x


def conv1x1_bn_pair(a, b, x):
    """Two conv1x1_bn modules that read the SAME feature (BiFPN's p4in_m / p4in_out, p5in_m / p5in_out,
    reference: models/fpns.py:366-372) as one launch: their prepared weights and BatchNorm terms are concatenated along
    the output channels (cached; rebuilt when either module's parameters change) and the two results are channel
    ranges of one pixel-major tensor, which the pyramid-node kernel reads in place through its leading dimension."""
    if a is _identity:
        return x, x
    pa, pb = prepare_conv(a, 'main', a[0], a[1]), prepare_conv(b, 'main', b[0], b[1])
    cache = a.__dict__.setdefault('_prep_cache', {})
    hit = cache.get('pair')
    if hit is None or hit[0][0] is not pa[0] or hit[0][1] is not pb[0]:
        with torch.no_grad():
            hit = ((pa[0], pb[0]), tuple(torch.cat([u, v]).contiguous() for u, v in zip(pa, pb)))
        cache['pair'] = hit
    w, scale, shift = hit[1]
    y = ops.conv2d(x, w, scale, shift, 1, 1, (0, 0, 0, 0), ops.ACT_NONE)
    na = pa[0].shape[0]
    return y[:, :na], y[:, na:]


class BiFPN3(nn.Module):
    '''
    One bidirectional pyramid layer over P3..P5 for three-level backbones (reference: models/fpns.py:315-354):
        P4m = fuse(p4in_m(P4in), up(p5in_4m(P5in)));  P3out = fuse(p3in_out(P3in), up(P4m));
        P4out = fuse(p4in_out(P4in), P4m, pool(P3out));  P5out = fuse(p5in_out(P5in), pool(P4out)).
    Every node is one launch of the fused node kernel; the 2x upsampling / max pooling are read through.
    '''
    def __init__(self, fpn_ch, fusion_method='linear', in_chs=None):
        super().__init__()
        if in_chs:
            assert len(in_chs) == 3
            self.p3in_out = conv1x1_bn(in_chs[0], fpn_ch)
            self.p4in_m = conv1x1_bn(in_chs[1], fpn_ch)
            self.p4in_out = conv1x1_bn(in_chs[1], fpn_ch)
            self.p5in_4m = conv1x1_bn(in_chs[2], fpn_ch)
            self.p5in_out = conv1x1_bn(in_chs[2], fpn_ch)
        else:
            self.p3in_out = self.p4in_m = self.p4in_out = self.p5in_4m = self.p5in_out = _identity
        if fusion_method != 'linear':
            raise NotImplementedError()
        self.fuse_4m = LinearFusion(num=2, channels=fpn_ch)
        self.fuse_3out = LinearFusion(num=2, channels=fpn_ch)
        self.fuse_4out = LinearFusion(num=3, channels=fpn_ch)
        self.fuse_5out = LinearFusion(num=2, channels=fpn_ch)

    def forward(self, features):
        P3in, P4in, P5in = features
        assert P3in.shape[2] == P4in.shape[2] * 2 == P5in.shape[2] * 4
        up, down = ops.FUSE_UP2X, ops.FUSE_POOL
        p5_4m, p5_out = conv1x1_bn_pair(self.p5in_4m, self.p5in_out, P5in)
        p4_m, p4_out = conv1x1_bn_pair(self.p4in_m, self.p4in_out, P4in)
        P4m = self.fuse_4m(p4_m, (p5_4m, up))
        P3out = self.fuse_3out(self.p3in_out(P3in), (P4m, up))
        P4out = self.fuse_4out(p4_out, P4m, (P3out, down))
        P5out = self.fuse_5out(p5_out, (P4out, down))
        return [P3out, P4out, P5out]


class BiFPN5(nn.Module):
    '''
    One bidirectional pyramid layer over P3..P7 (reference: models/fpns.py:357-418).  The nearest-2x
    upsampling of the top-down path and the max_pool2d(3,2,1) of the bottom-up path are never
    materialised: the fusion kernel reads the coarser / finer map through them.
    '''
    def __init__(self, fpn_ch, fusion_method='linear', in_chs=None):
        super().__init__()
        if in_chs:
            assert len(in_chs) == 5
            assert in_chs[3] == in_chs[4] == fpn_ch
            self.p3in_out = conv1x1_bn(in_chs[0], fpn_ch)
            self.p4in_m = conv1x1_bn(in_chs[1], fpn_ch)
            self.p4in_out = conv1x1_bn(in_chs[1], fpn_ch)
            self.p5in_m = conv1x1_bn(in_chs[2], fpn_ch)
            self.p5in_out = conv1x1_bn(in_chs[2], fpn_ch)
        else:
            self.p3in_out = self.p4in_m = self.p4in_out = self.p5in_m = self.p5in_out = _identity
        if fusion_method != 'linear':
            raise NotImplementedError()
        self.fuse_6m = LinearFusion(num=2, channels=fpn_ch)
        self.fuse_5m = LinearFusion(num=2, channels=fpn_ch)
        self.fuse_4m = LinearFusion(num=2, channels=fpn_ch)
        self.fuse_3out = LinearFusion(num=2, channels=fpn_ch)
        self.fuse_4out = LinearFusion(num=3, channels=fpn_ch)
        self.fuse_5out = LinearFusion(num=3, channels=fpn_ch)
        self.fuse_6out = LinearFusion(num=3, channels=fpn_ch)
        self.fuse_7out = LinearFusion(num=2, channels=fpn_ch)

    def forward(self, features):
        P3in, P4in, P5in, P6in, P7in = features
        assert P3in.shape[2] == P4in.shape[2] * 2 == P5in.shape[2] * 4 == P6in.shape[2] * 8 == P7in.shape[2] * 16
        up, down = ops.FUSE_UP2X, ops.FUSE_POOL
        p5_m, p5_out = conv1x1_bn_pair(self.p5in_m, self.p5in_out, P5in)
        p4_m, p4_out = conv1x1_bn_pair(self.p4in_m, self.p4in_out, P4in)
        P6m = self.fuse_6m(P6in, (P7in, up))
        P5m = self.fuse_5m(p5_m, (P6m, up))
        P4m = self.fuse_4m(p4_m, (P5m, up))
        P3out = self.fuse_3out(self.p3in_out(P3in), (P4m, up))
        P4out = self.fuse_4out(p4_out, P4m, (P3out, down))
        P5out = self.fuse_5out(p5_out, P5m, (P4out, down))
        P6out = self.fuse_6out(P6in, P6m, (P5out, down))
        P7out = self.fuse_7out(P7in, (P6out, down))
        return [P3out, P4out, P5out, P6out, P7out]


class LinearFusion(nn.Module):
    '''
    fused = sum(w_i * x_i), w = relu(weights) / (sum + 1e-4); out = BN(pointwise(depthwise(swish(fused))))
    (reference: models/fpns.py:421-439).  A feature may be given as (tensor, ops.FUSE_UP2X | ops.FUSE_POOL)
    to be read through nearest-2x upsampling / 3x3-stride-2 max pooling inside the fusion kernel.
    '''
    def __init__(self, num, channels):
        super().__init__()
        self.num = num
        self.weights = nn.Parameter(torch.ones(num), requires_grad=True)
        self.spconv_bn = SpconvBn(channels, swish=False)

    def forward(self, *features):
        assert isinstance(features, (list, tuple)) and len(features) == self.num
        tensors = [f[0] if isinstance(f, tuple) else f for f in features]
        modes = [f[1] if isinstance(f, tuple) else ops.FUSE_SAME for f in features]
        if self.spconv_bn.fusable():          # fusion + swish + depthwise + pointwise + BN: one launch
            return ops.sepconv_nodes([self.spconv_bn.node(tensors, modes, self.weights.detach())])[0]
        fused = ops.bifpn_fuse(tensors, modes, self.weights.detach())
        return self.spconv_bn(fused)
