"""Oracle (test infrastructure only): torch-CPU restatement of the YOLOv3-80 forward.

Follows, for a ``state_dict`` with the reference key names:
  ConvBnLeaky      models/modules.py:76-95   act(bn(conv(x))), BN eps 1e-5, LeakyReLU 0.1
  DarkBlock        models/modules.py:56-73   x + cbl_1(cbl_0(x))
  Darknet53        models/backbones.py:6-57  29-entry netlist, taps after 14 / 23 / 28
  YOLOv3FPN/Branch models/fpns.py:6-74       P5 first, 'process' 1x1 -> nearest upsample -> cat((pre, x))
  YOLOHead         models/rpns.py:8-45       1x1 conv + bias, channel = a*85 + {x,y,w,h,conf,cls...}
  YOLOLayer        models/detlayers/yolov3.py:30-69 (labels=None branch)
  OneStageBBox     models/general.py:44-87   concat levels along dim 1, order P3,P4,P5
Pinned against the imported reference by tests/golden/yolov3_*.npz.
"""
import torch
import torch.nn.functional as F

YOLO_ANCHORS = [[10, 13], [16, 30], [33, 23], [30, 61], [62, 45], [59, 119],
                [116, 90], [156, 198], [373, 326]]          # configs/yolov3_80.json:27-31
YOLO_ANCHOR_INDICES = [[0, 1, 2], [3, 4, 5], [6, 7, 8]]     # configs/yolov3_80.json:32
YOLO_STRIDES = (8, 16, 32)                                   # models/registry.py:24


def conv_bn_leaky(x, sd, p, stride=1):
    w = sd[p + '.conv.weight']
    k = w.shape[-1]
    y = F.conv2d(x, w, None, stride, (k - 1) // 2)
    y = F.batch_norm(y, sd[p + '.bn.running_mean'], sd[p + '.bn.running_var'],
                     sd[p + '.bn.weight'], sd[p + '.bn.bias'], False, 0.0, 1e-5)
    return F.leaky_relu(y, 0.1)


def dark_block(x, sd, p):
    return x + conv_bn_leaky(conv_bn_leaky(x, sd, p + '.cbl_0'), sd, p + '.cbl_1')


def darknet53(x, sd, p='backbone'):
    taps = {}
    for i in range(29):
        q = f'{p}.netlist.{i}'
        if q + '.conv.weight' in sd:            # plain CBL: index 0 is stride 1, the rest downsample
            x = conv_bn_leaky(x, sd, q, stride=1 if i == 0 else 2)
        else:
            x = dark_block(x, sd, q)
        if i in (14, 23, 28):
            taps[i] = x
    return [taps[14], taps[23], taps[28]]


def yolo_branch(x, sd, p, previous=None):
    if previous is not None:
        pre = conv_bn_leaky(previous, sd, p + '.process')
        pre = F.interpolate(pre, size=x.shape[2:4], mode='nearest')
        x = torch.cat((pre, x), dim=1)
    for j in range(4):
        x = conv_bn_leaky(x, sd, f'{p}.cbl_{j}')
    feature = conv_bn_leaky(x, sd, p + '.cbl_4')
    return conv_bn_leaky(feature, sd, p + '.cbl_5'), feature


def yolov3_fpn(feats, sd, p='fpn'):
    c3, c4, c5 = feats
    p5, to4 = yolo_branch(c5, sd, p + '.branch_P5')
    p4, to3 = yolo_branch(c4, sd, p + '.branch_P4', to4)
    p3, _ = yolo_branch(c3, sd, p + '.branch_P3', to3)
    return [p3, p4, p5]


def yolo_head(feats, sd, p='rpn', n_anch=3, n_cls=80):
    """Returns the raw conv outputs [B, A*(5+C), H, W] per level (pre-view)."""
    return [F.conv2d(f, sd[f'{p}.heads.conv_{i}.weight'], sd[f'{p}.heads.conv_{i}.bias'])
            for i, f in enumerate(feats)]


def yolo_decode(raw, level, n_cls=80, anchors=YOLO_ANCHORS, indices=YOLO_ANCHOR_INDICES,
                strides=YOLO_STRIDES):
    """raw [B, A*(5+C), H, W] -> bbox [B,A*H*W,4], class_idx [B,AHW] i64, score [B,AHW]."""
    nB, _, nH, nW = raw.shape
    anch = torch.tensor(anchors, dtype=torch.float32)[indices[level]]
    nA = anch.shape[0]
    stride = strides[level]
    t = raw.view(nB, nA, 5 + n_cls, nH, nW).permute(0, 1, 3, 4, 2)
    xywh = t[..., 0:4].clone().contiguous()
    ys = torch.arange(nH, dtype=torch.float32).view(1, 1, nH, 1)
    xs = torch.arange(nW, dtype=torch.float32).view(1, 1, 1, nW)
    xywh[..., 0] = (torch.sigmoid(xywh[..., 0]) + xs) * stride
    xywh[..., 1] = (torch.sigmoid(xywh[..., 1]) + ys) * stride
    xywh[..., 2:4] = torch.exp(xywh[..., 2:4]) * anch.view(1, nA, 1, 1, 2)
    conf = torch.sigmoid(t[..., 4:5])
    cls_score, cls_idx = torch.max(torch.sigmoid(t[..., 5:]), dim=-1, keepdim=True)
    score = conf * cls_score
    n = nA * nH * nW
    return xywh.view(nB, n, 4), cls_idx.reshape(nB, n), score.reshape(nB, n)


def forward_features(x, sd):
    return yolov3_fpn(darknet53(x, sd), sd)


def forward(x, sd, return_raw=False):
    """x [B,3,H,W] -> (bbox [B,N,4], class_idx [B,N], score [B,N]), levels concatenated P3,P4,P5."""
    raws = yolo_head(forward_features(x, sd), sd)
    outs = [yolo_decode(r, i) for i, r in enumerate(raws)]
    res = tuple(torch.cat([o[j] for o in outs], dim=1) for j in range(3))
    return res + (raws,) if return_raw else res
