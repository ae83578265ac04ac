"""Oracle (test infrastructure only): torch-CPU restatement of the EfficientDet-D1 and
D1-FCOS2-ATSS forward for a ``state_dict`` with the reference key names.

Follows
  EfficientNet-B1 trunk   external/efficientnet/model.py:71-98 (MBConvBlock.forward), :118-140 (block list),
                          external/efficientnet/utils.py:59-79 (round_filters/repeats), :122-145 (static SAME
                          pad from image_size 240), :166 (b1 = width 1.0, depth 1.1), :258-263 (stage strings)
  EfNetBackbone           models/backbones.py:166-232 (taps where width changes; C6/C7 'maxpool' | 'conv')
  BiFPN5 / LinearFusion   models/fpns.py:357-448
  SeparableConv2d         models/modules.py:5-21
  EfDetHead               models/rpns.py:121-205
  RetinaLayer / FCOS_ATSS oracle/decoders.py
Pinned against the imported reference by tests/golden/{efficientdet_d1,d1_fcs2_atss}_b1_256.npz.
"""
import math

import torch
import torch.nn.functional as F

from . import decoders

BN_EPS = 1e-3
STAGES_B0 = [  # (repeat, kernel, stride, expand, in, out) -- external/efficientnet/utils.py:258-263, se_ratio 0.25
    (1, 3, 1, 1, 32, 16), (2, 3, 2, 6, 16, 24), (2, 5, 2, 6, 24, 40), (3, 3, 2, 6, 40, 80),
    (3, 5, 1, 6, 80, 112), (4, 5, 2, 6, 112, 192), (1, 3, 1, 6, 192, 320)]
COEFFS = {'efficientnet-b0': (1.0, 1.0), 'efficientnet-b1': (1.0, 1.1), 'efficientnet-b2': (1.1, 1.2)}
STATIC_IMAGE_SIZE = {'efficientnet-b0': 224, 'efficientnet-b1': 240, 'efficientnet-b2': 260}
STRIDES = (8, 16, 32, 64, 128)


def _round_filters(f, width, divisor=8):
    f = f * width
    new = max(divisor, int(f + divisor / 2) // divisor * divisor)
    if new < 0.9 * f:
        new += divisor
    return int(new)


def block_table(name='efficientnet-b1'):
    """[(kernel, stride, expand, in, out, se_channels)] per MBConv block."""
    width, depth = COEFFS[name]
    out = []
    for rep, k, s, e, cin, cout in STAGES_B0:
        cin, cout = _round_filters(cin, width), _round_filters(cout, width)
        rep = int(math.ceil(depth * rep))
        out.append((k, s, e, cin, cout, max(1, int(cin * 0.25))))
        for _ in range(rep - 1):
            out.append((k, 1, e, cout, cout, max(1, int(cout * 0.25))))
    return out


def same_pad(k, s, image_size):
    """(left, right) == (top, bottom) static TF-SAME pad computed from the model's nominal image size."""
    o = math.ceil(image_size / s)
    pad = max((o - 1) * s + k - image_size, 0)
    return pad // 2, pad - pad // 2


def _bn(x, sd, p, eps=BN_EPS):
    return F.batch_norm(x, sd[p + '.running_mean'], sd[p + '.running_var'], sd[p + '.weight'], sd[p + '.bias'],
                        False, 0.0, eps)


def _swish(x):
    return x * torch.sigmoid(x)


def _same_conv(x, w, b, k, s, groups, image_size):
    lo, hi = same_pad(k, s, image_size)
    if lo or hi:
        x = F.pad(x, [lo, hi, lo, hi])
    return F.conv2d(x, w, b, s, 0, 1, groups)


def mbconv(x, sd, p, k, s, e, cin, cout, image_size):
    inp = x
    if e != 1:
        x = _swish(_bn(_same_conv(x, sd[p + '._expand_conv.weight'], None, 1, 1, 1, image_size), sd, p + '._bn0'))
    c = x.shape[1]
    x = _swish(_bn(_same_conv(x, sd[p + '._depthwise_conv.weight'], None, k, s, c, image_size), sd, p + '._bn1'))
    sq = F.adaptive_avg_pool2d(x, 1)
    sq = F.conv2d(_swish(F.conv2d(sq, sd[p + '._se_reduce.weight'], sd[p + '._se_reduce.bias'])),
                  sd[p + '._se_expand.weight'], sd[p + '._se_expand.bias'])
    x = torch.sigmoid(sq) * x
    x = _bn(F.conv2d(x, sd[p + '._project_conv.weight']), sd, p + '._bn2')
    if s == 1 and cin == cout:
        x = x + inp                                   # drop_connect is identity in eval
    return x


def backbone(x, sd, name='efficientnet-b1', c6c7='maxpool', p='backbone'):
    isz = STATIC_IMAGE_SIZE[name]
    x = _swish(_bn(_same_conv(x, sd[p + '.model._conv_stem.weight'], None, 3, 2, 1, isz), sd, p + '.model._bn0'))
    feats = []
    for i, (k, s, e, cin, cout, _) in enumerate(block_table(name)):
        y = mbconv(x, sd, f'{p}.model._blocks.{i}', k, s, e, cin, cout, isz)
        if y.shape[-1] != x.shape[-1]:
            feats.append(x)
        x = y
    feats.append(x)
    c3, c4, c5 = feats[2], feats[3], feats[4]
    if c6c7 is None:                                  # three-level backbone (models/backbones.py:174-177)
        return [c3, c4, c5]
    pad6 = 0 if c6c7 == 'maxpool' else 1
    c6 = F.conv2d(c5, sd[p + '.c5_to_c6.0.weight'], sd[p + '.c5_to_c6.0.bias'], 1, pad6)
    c6 = F.max_pool2d(_bn(c6, sd, p + '.c5_to_c6.1'), 3, 2, 1)
    if c6c7 == 'maxpool':
        c7 = F.max_pool2d(c6, 3, 2, 1)
    else:
        c7 = F.conv2d(c6, sd[p + '.c6_to_c7.0.weight'], sd[p + '.c6_to_c7.0.bias'], 1, 1)
        c7 = F.max_pool2d(_bn(c7, sd, p + '.c6_to_c7.1'), 3, 2, 1)
    return [c3, c4, c5, c6, c7]


def sepconv(x, sd, p, pad=1):
    x = F.conv2d(x, sd[p + '.depthwise.weight'], None, 1, pad, 1, x.shape[1])
    return F.conv2d(x, sd[p + '.pointwise.weight'], sd[p + '.pointwise.bias'])


def fusion(feats, sd, p):
    w = F.relu(sd[p + '.weights'])
    w = w / (w.sum() + 0.0001)
    fused = sum([wi * f for wi, f in zip(w, feats)])
    return _bn(sepconv(_swish(fused), sd, p + '.spconv_bn.0'), sd, p + '.spconv_bn.1')


def _proj(x, sd, p):
    return _bn(F.conv2d(x, sd[p + '.0.weight'], sd[p + '.0.bias']), sd, p + '.1') if p + '.0.weight' in sd else x


def bifpn5(feats, sd, p):
    p3, p4, p5, p6, p7 = feats
    up = lambda t: F.interpolate(t, scale_factor=(2, 2), mode='nearest')    # noqa: E731
    down = lambda t: F.max_pool2d(t, kernel_size=3, stride=2, padding=1)      # noqa: E731
    p6m = fusion([p6, up(p7)], sd, p + '.fuse_6m')
    p5m = fusion([_proj(p5, sd, p + '.p5in_m'), up(p6m)], sd, p + '.fuse_5m')
    p4m = fusion([_proj(p4, sd, p + '.p4in_m'), up(p5m)], sd, p + '.fuse_4m')
    p3o = fusion([_proj(p3, sd, p + '.p3in_out'), up(p4m)], sd, p + '.fuse_3out')
    p4o = fusion([_proj(p4, sd, p + '.p4in_out'), p4m, down(p3o)], sd, p + '.fuse_4out')
    p5o = fusion([_proj(p5, sd, p + '.p5in_out'), p5m, down(p4o)], sd, p + '.fuse_5out')
    p6o = fusion([p6, p6m, down(p5o)], sd, p + '.fuse_6out')
    p7o = fusion([p7, down(p6o)], sd, p + '.fuse_7out')
    return [p3o, p4o, p5o, p6o, p7o]


def bifpn3(feats, sd, p):
    """BiFPN3.forward, models/fpns.py:336-354."""
    p3, p4, p5 = feats
    up = lambda t: F.interpolate(t, scale_factor=(2, 2), mode='nearest')    # noqa: E731
    down = lambda t: F.max_pool2d(t, kernel_size=3, stride=2, padding=1)      # noqa: E731
    p4m = fusion([_proj(p4, sd, p + '.p4in_m'), up(_proj(p5, sd, p + '.p5in_4m'))], sd, p + '.fuse_4m')
    p3o = fusion([_proj(p3, sd, p + '.p3in_out'), up(p4m)], sd, p + '.fuse_3out')
    p4o = fusion([_proj(p4, sd, p + '.p4in_out'), p4m, down(p3o)], sd, p + '.fuse_4out')
    p5o = fusion([_proj(p5, sd, p + '.p5in_out'), down(p4o)], sd, p + '.fuse_5out')
    return [p3o, p4o, p5o]


def bifpn(feats, sd, repeat=4, p='fpn'):
    layer = bifpn3 if len(feats) == 3 else bifpn5
    for i in range(repeat):
        feats = layer(feats, sd, f'{p}.{i}')
    return feats


def head(feats, sd, repeat=3, p='rpn'):
    """Per level: (class conv output [B, A*K, H, W], bbox conv output [B, A*4, H, W])."""
    outs = []
    for lvl, x in enumerate(feats):
        res = []
        for net in ('class_nets', 'bbox_nets'):
            t = x
            for r in range(repeat):
                q = f'{p}.{net}.{lvl}.{r}'
                t = _swish(_bn(sepconv(t, sd, q + '.0'), sd, q + '.1'))
            q = f'{p}.{net}.{lvl}.{repeat}'
            if q + '.pointwise.weight' in sd:
                t = sepconv(t, sd, q)
            else:
                t = F.conv2d(t, sd[q + '.weight'], sd[q + '.bias'], 1, 1)
            res.append(t)
        outs.append(tuple(res))
    return outs


def head_with_center(feats, sd, repeat=3, p='rpn'):
    """EfDetHead_wCenter (models/rpns.py:232-312): per level (class [B,K,H,W], bbox [B,4,H,W], center [B,1,H,W])."""
    outs = []
    for lvl, x in enumerate(feats):
        def tower(net, t, n):
            for r in range(n):
                q = f'{p}.{net}.{lvl}.{r}'
                t = _swish(_bn(sepconv(t, sd, q + '.0'), sd, q + '.1'))
            return t
        c = tower('class_nets', x, repeat)
        c = F.conv2d(c, sd[f'{p}.class_nets.{lvl}.{repeat}.weight'], sd[f'{p}.class_nets.{lvl}.{repeat}.bias'], 1, 1)
        bf = tower('bbox_nets', x, repeat)
        b = F.conv2d(bf, sd[f'{p}.bbox_lasts.{lvl}.weight'], sd[f'{p}.bbox_lasts.{lvl}.bias'], 1, 1)
        ct = tower('center_nets', bf, 1)
        ct = F.conv2d(ct, sd[f'{p}.center_nets.{lvl}.1.weight'], sd[f'{p}.center_nets.{lvl}.1.bias'], 1, 1)
        outs.append((c, b, ct))
    return outs


def raw_dicts(head_outs, n_anchor, n_cls, enable_conf):
    raws = []
    for cls, box in head_outs:
        nB, _, nH, nW = box.shape
        if n_anchor >= 2:
            box = box.view(nB, n_anchor, -1, nH, nW).permute(0, 1, 3, 4, 2)
            cls = cls.view(nB, n_anchor, -1, nH, nW).permute(0, 1, 3, 4, 2)
        else:
            box, cls = box.permute(0, 2, 3, 1), cls.permute(0, 2, 3, 1)
        raws.append({'bbox': box, 'conf': cls[..., 0:1], 'class': cls[..., 1:]} if enable_conf
                    else {'bbox': box, 'class': cls})
    return raws


# configs/d1_yv3.json:31-41 (anchor_indices are consecutive triples)
D1_YV3_ANCHORS = [[12.6, 13.2], [23.5, 38.1], [57.3, 32.3], [42.9, 75.5], [106.6, 61.2], [60.4, 123.5],
                  [84.5, 191.6], [131.9, 123.9], [212.4, 85.6], [125.4, 278.9], [179.6, 196.4], [347.0, 107.1],
                  [272.3, 199.2], [238.8, 321.5], [373.1, 258.9]]


def class_margin(cls):
    """How well defined the class id of every candidate is, [B, n] in the decoders' flatten order; cls [..., n_cls] logits.
    The gap between the two largest class probabilities p1 - p2 where p1 >= 0.05; below that, the gap RELATIVE to p1, scaled so
    that a relative gap of 4e-4 reads as 2e-5: round-off moves a probability by eps * p (1 - p) for a logit error eps, i.e. by a
    fixed amount near p ~ 0.5 and by a fixed FRACTION where p is tiny (background candidates: p ~ 1e-9, gaps ~ 1e-10 absolute
    and still a hundred times wider than round-off).  Class ids are compared where this exceeds 2e-5."""
    top2 = torch.sigmoid(cls).reshape(cls.shape[0], -1, cls.shape[-1]).topk(2, dim=-1).values
    return (top2[..., 0] - top2[..., 1]) * torch.clamp(0.05 / top2[..., 0].clamp_min(1e-38), min=1.0)


def forward(x, sd, config, with_margin=False):
    """with_margin: also return class_margin of every candidate, [B,N].
    config in {'efficientdet-d1', 'd1_fcs2_atss', 'd1_fcs2', 'd1_fcs2_p3', 'd1_fcs', 'd1_yv3'} -> (bbox [B,N,4], class_idx [B,N],
    score [B,N]).  d1_fcs2 is d1_fcs2_atss at inference (models/detlayers/fcos2.py:24-69 == :222-251)."""
    img = tuple(x.shape[2:4])
    if config == 'd1_fcs':          # EfDetHead_wCenter + FCOSLayer (models/detlayers/fcos.py:21-68)
        feats = bifpn(backbone(x, sd, c6c7='maxpool'), sd)
        outs = []
        for lvl, (c, b, ct) in enumerate(head_with_center(feats, sd)):
            raw = {'bbox': b.permute(0, 2, 3, 1), 'conf': ct.permute(0, 2, 3, 1), 'class': c.permute(0, 2, 3, 1)}
            outs.append(decoders.fcos_decode(raw, img, STRIDES[lvl]) + (class_margin(raw['class']),))
        return tuple(torch.cat([o[j] for o in outs], dim=1) for j in range(4 if with_margin else 3))
    if config == 'd1_yv3':          # EfDetHead (3 anchors, conf = class channel 0) + YOLOLayer on 5 levels
        feats = bifpn(backbone(x, sd, c6c7='maxpool'), sd)
        outs = []
        for lvl, raw in enumerate(raw_dicts(head(feats, sd), 3, 80, True)):
            anch = torch.tensor(D1_YV3_ANCHORS[3 * lvl:3 * lvl + 3], dtype=torch.float32)
            outs.append(decoders.yolo_decode_raw(raw, STRIDES[lvl], anch) + (class_margin(raw['class']),))
        return tuple(torch.cat([o[j] for o in outs], dim=1) for j in range(4 if with_margin else 3))
    atss = config in ('d1_fcs2_atss', 'd1_fcs2', 'd1_fcs2_p3')
    c6c7 = None if config == 'd1_fcs2_p3' else ('conv' if atss else 'maxpool')
    feats = bifpn(backbone(x, sd, c6c7=c6c7), sd)
    raws = raw_dicts(head(feats, sd), 1 if atss else 9, 80, atss)
    outs = []
    for lvl, raw in enumerate(raws):
        if atss:
            outs.append(decoders.fcos_decode(raw, img, STRIDES[lvl]) + (class_margin(raw['class']),))
        else:
            outs.append(decoders.retina_decode(raw, img, STRIDES[lvl], decoders.retina_anchors(STRIDES[lvl])) + (class_margin(raw['class']),))
    return tuple(torch.cat([o[j] for o in outs], dim=1) for j in range(4 if with_margin else 3))
