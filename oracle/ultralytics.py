"""Oracle (test infrastructure only): torch-CPU restatement of the Ultralytics (YOLOv5) trunk and pyramid of the
reference under its YOLOv3 head and decode (configs/u5m_yv3.json).

Follows, for a ``state_dict`` with the reference key names:
  Conv            external/ultralytics/common.py:12-25   act(bn(conv(x))), BatchNorm eps 1e-5 (nn.BatchNorm2d default), LeakyReLU 0.1
  Bottleneck      common.py:28-37     x + cv2(cv1(x)) when shortcut and c1 == c2
  BottleneckCSP   common.py:40-56     cv4(act(bn(cat(cv3(m(cv1(x))), cv2(x)))))
  SPP             common.py:59-70     cv2(cat([x, maxpool5(x), maxpool9(x), maxpool13(x)])), x = cv1(input)
  Focus           common.py:79-86     conv(cat([x[..., ::2, ::2], x[..., 1::2, ::2], x[..., ::2, 1::2], x[..., 1::2, 1::2]]))
  UltralyticsBackbone  models/backbones.py:60-113   features where the next module halves the map, last three
  UltralyticsFPN       models/fpns.py:77-107        top-down, nearest 2x + cat([up, c]) + Conv 1x1 + CSP (no shortcut)
  YOLOHead / YOLOLayer / concat: oracle/yolov3.py (same code path as yolov3_80)
Pinned against the imported reference by tests/golden/u5m_yv3_b1_256.npz.
"""
import torch
import torch.nn.functional as F

from . import yolov3 as oy


def conv(x, sd, p, stride=1, act=True):
    w = sd[p + '.conv.weight']
    k = w.shape[-1]
    y = F.conv2d(x, w, None, stride, k // 2)
    y = F.batch_norm(y, sd[p + '.bn.running_mean'], sd[p + '.bn.running_var'], sd[p + '.bn.weight'], sd[p + '.bn.bias'],
                     False, 0.0, 1e-5)
    return F.leaky_relu(y, 0.1) if act else y


def bottleneck(x, sd, p, shortcut=True):
    y = conv(conv(x, sd, p + '.cv1'), sd, p + '.cv2')
    return x + y if shortcut and x.shape[1] == y.shape[1] else y


def _count(sd, prefix):
    n = 0
    while f'{prefix}.{n}.cv1.conv.weight' in sd:
        n += 1
    return n


def bottleneck_csp(x, sd, p, shortcut=True):
    y = conv(x, sd, p + '.cv1')
    for i in range(_count(sd, p + '.m')):
        y = bottleneck(y, sd, f'{p}.m.{i}', shortcut)
    y1 = F.conv2d(y, sd[p + '.cv3.weight'])
    y2 = F.conv2d(x, sd[p + '.cv2.weight'])
    z = torch.cat((y1, y2), dim=1)
    z = F.batch_norm(z, sd[p + '.bn.running_mean'], sd[p + '.bn.running_var'], sd[p + '.bn.weight'], sd[p + '.bn.bias'],
                     False, 0.0, 1e-5)
    return conv(F.leaky_relu(z, 0.1), sd, p + '.cv4')


def spp(x, sd, p, ks=(5, 9, 13)):
    x = conv(x, sd, p + '.cv1')
    return conv(torch.cat([x] + [F.max_pool2d(x, k, 1, k // 2) for k in ks], 1), sd, p + '.cv2')


def focus(x, sd, p):
    return conv(torch.cat([x[..., ::2, ::2], x[..., 1::2, ::2], x[..., ::2, 1::2], x[..., 1::2, 1::2]], 1), sd, p + '.conv')


def backbone(x, sd, p='backbone'):
    q = p + '.netlist'
    feats = []

    def step(y):
        nonlocal x
        if y.shape[2:4] != x.shape[2:4]:
            feats.append(x)
        x = y
    step(focus(x, sd, q + '.0'))
    step(conv(x, sd, q + '.1', stride=2))
    y = x
    for i in range(_count(sd, q + '.2')):
        y = bottleneck(y, sd, f'{q}.2.{i}')
    step(y)
    step(conv(x, sd, q + '.3', stride=2))
    step(bottleneck_csp(x, sd, q + '.4'))
    step(conv(x, sd, q + '.5', stride=2))
    step(bottleneck_csp(x, sd, q + '.6'))
    step(conv(x, sd, q + '.7', stride=2))
    step(spp(x, sd, q + '.8'))
    step(bottleneck_csp(x, sd, q + '.9'))
    feats.append(x)
    assert len(feats) == 6
    return feats[3:]


def fpn(feats, sd, p='fpn'):
    c3, c4, c5 = feats
    p5 = bottleneck_csp(c5, sd, p + '.to_p5', shortcut=False)
    x = torch.cat([F.interpolate(p5, scale_factor=(2, 2), mode='nearest'), c4], dim=1)
    p4 = bottleneck_csp(conv(x, sd, p + '.to_p4.0'), sd, p + '.to_p4.1', shortcut=False)
    x = torch.cat([F.interpolate(p4, scale_factor=(2, 2), mode='nearest'), c3], dim=1)
    p3 = bottleneck_csp(conv(x, sd, p + '.to_p3.0'), sd, p + '.to_p3.1', shortcut=False)
    return [p3, p4, p5]


def forward_features(x, sd):
    return fpn(backbone(x, sd), sd)


def forward(x, sd, return_raw=False):
    """u5m_yv3: x [B,3,H,W] -> (bbox [B,N,4], class_idx [B,N], score [B,N]), levels concatenated P3,P4,P5."""
    raws = oy.yolo_head(forward_features(x, sd), sd)
    outs = [oy.yolo_decode(r, i) for i, r in enumerate(raws)]
    res = tuple(torch.cat([o[j] for o in outs], dim=1) for j in range(3))
    return res + (raws,) if return_raw else res
