"""Deterministic synthetic weights and inputs (no network, no checkpoints).

The reference needs pretrained files that do not exist here
(`models/registry.py:15` loads weights/dark53_imgnet.pth; `models/backbones.py:171`
downloads EfficientNet weights), so every model in this repo, the oracle and the
golden-vector generator are filled from this one generator.  A tensor is a pure
function of its ``state_dict`` key and shape: numpy PCG64 seeded from SHA-256 of
the key, so the reference model in the build container and the HIP model on the
GPU box get bit-identical parameters without shipping them.

Recipe: conv weights zero-mean normal with a
variance-preserving gain, BN gamma~U(.5,1.5), beta~N(0,.1), mean~N(0,.1),
var~U(.5,1.5); residual-branch BN gamma damped so 23 stacked DarkBlocks do not
blow up; head convs calibrated so objectness/class logits are long-tailed
(thousands of candidates >= 0.005, classes well spread) instead of the degenerate
all-0.25 scores plain random init gives.
"""
import hashlib

import numpy as np
import torch

# rms of the FPN output feeding each YOLO head conv, measured once with the
# recipe below (oracle forward, 512x512 uniform input); used to set head gains.
_YOLO_HEAD_FEATURE_RMS = {0: 1.14, 1: 1.61, 2: 2.64}
# the same for the Ultralytics (YOLOv5-m) trunk + pyramid under the YOLO head (u5m_yv3 / u5m_fcs2), 256x256 'rects' input
_U5M_HEAD_FEATURE_RMS = {0: 1.45, 1: 2.03, 2: 2.31}


def _rng(key: str) -> np.random.Generator:
    seed = int.from_bytes(hashlib.sha256(key.encode()).digest()[:8], 'little')
    return np.random.Generator(np.random.PCG64(seed))


def _normal(key, shape, std=1.0, mean=0.0):
    a = _rng(key).standard_normal(size=shape, dtype=np.float32)
    return (a * np.float32(std) + np.float32(mean)).astype(np.float32)


def _uniform(key, shape, lo, hi):
    a = _rng(key).random(size=shape, dtype=np.float32)
    return (a * np.float32(hi - lo) + np.float32(lo)).astype(np.float32)


# Calibrated YOLO head (configs that ship per-channel head statistics, oracle/calibrate_yolo_head.py -> calib/<config>.npz):
# (std, mean) of each kind of head logit over images and cells.  Objectness is wide and far below zero, so the score
# sigmoid(conf) * max_c sigmoid(cls_c) has a long upper tail instead of a narrow band: at 640^2 some 700-900 of the 25 200
# candidates pass 0.005 (the top-512 cut applies; ~370 detections in ~45 classes), ~400 pass 0.05 and ~120 pass 0.5 (~85
# detections in ~15 classes), and the winning class changes from cell to cell.  The spread is bought with gain on the
# SPATIAL variation of the pyramid features (a third of their magnitude), which amplifies float32 round-off of the features
# by the same factor: 4.0 keeps the score error of a float32 forward near 1e-5 (6.4 gave 3e-5 through the F(4x4) layers).
# Class logits stay below 5 (the decode kernel's collision-exact path for saturated classes is for trained weights).
_YOLO_TARGETS = {'xy': (1.0, 0.0), 'wh': (0.35, 0.0), 'conf': (4.0, -13.0), 'class': (2.0, -3.5)}


def yolo_head_unit_weight(key, shape):
    """Head conv weight before any gain: N(0, 1 / fan_in) per element, a pure function of the key."""
    fan_in = shape[1] * shape[2] * shape[3]
    return (_normal(key, shape) / np.float32(np.sqrt(fan_in))).astype(np.float32)


def _yolo_head(key, shape, n_cls=80, feature_rms=None, calib=None):
    """rpn.heads.conv_{i}.{weight,bias}: row o = a*(5+n_cls) + c  (models/rpns.py:27-33)."""
    feature_rms = _YOLO_HEAD_FEATURE_RMS if feature_rms is None else feature_rms
    level = int(key.split('conv_')[1].split('.')[0])
    rows = shape[0]
    c = np.arange(rows) % (5 + n_cls)
    module = key.rsplit('.', 1)[0]
    rowstd = (calib or {}).get('__rowstd__/' + module)
    if rowstd is not None and rowstd.shape[0] == rows:
        # every output channel normalised with its measured unit-gain statistics, then given its kind's target
        kind = np.where(c < 2, 0, np.where(c < 4, 1, np.where(c == 4, 2, 3)))
        tstd = np.array([_YOLO_TARGETS[k][0] for k in ('xy', 'wh', 'conf', 'class')], np.float32)[kind]
        tmean = np.array([_YOLO_TARGETS[k][1] for k in ('xy', 'wh', 'conf', 'class')], np.float32)[kind]
        gain = (tstd / rowstd.astype(np.float32)).astype(np.float32)
        if key.endswith('.bias'):
            rowmean = calib['__rowmean__/' + module].astype(np.float32)
            return (tmean - gain * rowmean + _normal(key, (rows,), std=0.05)).astype(np.float32)
        return yolo_head_unit_weight(key, shape) * gain.reshape(-1, 1, 1, 1)
    # target logit std per row kind: xy 1.0, wh 0.35, conf 2.5, class 2.0
    tgt = np.where(c < 2, 1.0, np.where(c < 4, 0.35, np.where(c == 4, 2.5, 2.0))).astype(np.float32)
    if key.endswith('.bias'):
        bias = np.where(c < 4, 0.0, np.where(c == 4, -5.0, -4.0)).astype(np.float32)
        return bias + _normal(key, (rows,), std=0.05)
    fan_in = shape[1] * shape[2] * shape[3]
    rms = feature_rms.get(level, 1.0)
    w = _normal(key, shape)
    return w * (tgt / (np.sqrt(fan_in) * rms)).reshape(-1, 1, 1, 1).astype(np.float32)


# ---------------------------------------------------------------- EfficientDet family (EfficientNet-B1 + BiFPN + EfDetHead)
# These nets are ~110 layers deep.  Through conv -> BN -> swish a float32 round-off perturbation grows by
# sqrt(E[swish'(z)^2] Var[z] / Var[swish(z)]) per non-linearity: 1.10 for z ~ N(0,1), i.e. exponentially with depth
# (x70 over the trunk with the SURVEY 8d recipe: the float32 reference itself then sits 3e-4 from a float64
# evaluation, outside north_star's 1e-4).  The recipe below keeps every layer non-linear but in the mildly
# non-linear regime (z ~ N(1, 0.6^2): growth 1.01 per layer) and damps the residual branches like the Darknet
# recipe does, so round-off accumulates additively (sqrt(depth) x 1.4e-7) and the 1e-4 gate is meaningful.
_PRE_SWISH_GAMMA = (0.4, 0.8)
_PRE_SWISH_BETA = (1.0, 0.25)        # mean, std
_RESIDUAL_GAMMA = (0.1, 0.3)
# recipe='stiff': a second, deliberately ill-conditioned parameter set for the same networks (what trained weights
# behave like: pre-activations centred on the swish's curved part, residual branches barely damped), where float32
# round-off DOES grow with depth.  No absolute gate can hold there; tests use it with the relative criterion
# "the HIP path is as close to a float64 evaluation as the float32 CPU reference is" (tests/test_gpu_model.py).
_STIFF = {'pre_swish_gamma': (0.9, 1.5), 'pre_swish_beta': (0.0, 0.25), 'residual_gamma': (0.5, 1.0)}

# Final-layer targets (long-tailed scores, well-spread classes).  Logit std / bias per output kind; the
# measured spread of each final layer's output under unit gain is part of the calibration file
# (oracle/calibrate_bn.py, key '__std__/<module>'), so the targets hold whatever the tower statistics are.
# Round 6 (VERDICT r05 #2a): with the round-2 targets ~all candidates passed 0.005 AND more than 512 passed 0.5, so the three
# post-processing settings of the configs were one setting.  A trained detector's scores are an OBJECTNESS-like factor (shared by
# the classes of a candidate, far below zero almost everywhere) times a class preference; the recipe now has that structure:
#   * heads with a conf / centerness channel (FCOS, FCOS-ATSS, YOLO on the EfDetHead): that channel gets 'conf' / 'center';
#   * RetinaNet (class logits only): every class row of anchor a also carries the anchor's objectness direction o_a = the
#     normalised average of its 80 unit rows, with gain 'object' (std, bias) -- so logit(a, k) = g_C u(a, k) + s_O o(a) + b;
#   * `level_shift` / `level_shift_conf`: added to the objectness bias by pyramid level counted from the COARSEST (index 0): the
#     fine levels hold 90 % of the candidates and are pushed below every threshold, as background is in a trained net.
# Chosen with tools/r06/effdet_targets.py at 640^2: > 512 candidates pass 0.005 (the top-512 cut applies), < 512 pass 0.05,
# 50-150 pass 0.5, >= 30 classes among the detections.
_EFDET_TARGETS = {'class': (1.6, -1.0), 'conf': (3.0, -2.5), 'center': (3.0, -2.5),
                  'class_only': (1.2, -3.0), 'object': (3.2, -4.3),   # RetinaNet: score = max over 80 classes, no objectness factor
                  'level_shift': (0.0, 0.0, -6.0, -12.0, -16.0),            # RetinaNet: 225 / 900 / 3 600 / 14 400 / 57 600 candidates at 640^2
                  'level_shift_conf': (0.0, 0.0, -3.0, -8.0, -14.0),           # one candidate per cell: 25 / 100 / 400 / 1 600 / 6 400
                  'ltrb': (0.4, 1.0),                               # anchor-free: log-distances to the four sides
                  'anchor_xy': (0.03, 0.0), 'anchor_wh': (0.3, 0.0),   # RetinaNet: offsets in units of the anchor size
                  'cell_xy': (1.0, 0.0), 'cell_wh': (0.3, 0.0)}        # YOLO: sigmoid offsets inside the cell
_N_CLS = 80          # every configuration of the family has 80 classes (general.num_class)


def _efdet_last_kind(key):
    """'class' | 'bbox' | 'center' for the final layers of EfDetHead / EfDetHead_wCenter (models/rpns.py:139-160,
    245-266): rpn.{class,bbox}_nets.{lvl}.3[.pointwise|.depthwise].*, rpn.bbox_lasts.{lvl}.*, rpn.center_nets.{lvl}.1.*;
    None for every other key."""
    parts = key.split('.')
    if key.startswith(('rpn.class_nets.', 'rpn.bbox_nets.')) and len(parts) >= 5 and parts[3] == '3':
        return 'class' if parts[1] == 'class_nets' else 'bbox'
    if key.startswith('rpn.bbox_lasts.'):
        return 'bbox'
    if key.startswith('rpn.center_nets.') and len(parts) == 5 and parts[3] == '1':
        return 'center'
    return None


def _efdet_row_targets(kind, rows, from_coarsest=0):
    """Per-output-channel (logit std, bias).  With `enable_conf` the class conv has A*(1+80) channels and channel
    a*81 is the objectness / centerness logit (models/rpns.py:186-195); an anchor-free bbox conv (4 channels) holds
    log-distances to the four sides (models/detlayers/fcos2.py:222-251), biased so boxes span a few strides.
    from_coarsest: the pyramid level of the layer counted from the coarsest one (the objectness bias is shifted by it)."""
    std, bias = np.empty(rows, np.float32), np.empty(rows, np.float32)
    ls = _EFDET_TARGETS['level_shift' if (kind == 'class' and rows == 9 * _N_CLS) else 'level_shift_conf']
    shift = np.float32(ls[min(from_coarsest, len(ls) - 1)])
    if kind in ('class', 'conf', 'center', 'ltrb'):
        std[:], bias[:] = _EFDET_TARGETS[kind]
    if kind == 'center':
        bias += shift
    if kind == 'class' and rows == 9 * _N_CLS:
        std[:], bias[:] = _EFDET_TARGETS['class_only']
        bias += np.float32(_EFDET_TARGETS['object'][1]) + shift        # the objectness part's bias (its gain: _efdet_last)
    if kind == 'class' and rows % (_N_CLS + 1) == 0:
        conf = np.arange(rows) % (_N_CLS + 1) == 0
        std[conf], bias[conf] = _EFDET_TARGETS['conf']
        bias[conf] += shift
    if kind == 'bbox':
        # 4 channels: FCOS (models/detlayers/fcos2.py:222-251); 9 anchors: RetinaNet, cx = acx + tx * aw with anchors up
        # to 1 149 px (models/detlayers/retinanet.py:21-28,63-70) -- a centre offset of a few percent of the anchor keeps
        # float32 round-off of the box inside 1e-4 + 1e-4 |v|; otherwise the YOLO decode (models/detlayers/yolov3.py:41-47)
        style = 'ltrb' if rows == 4 else None
        if style:
            std[:], bias[:] = _EFDET_TARGETS['ltrb']
        else:
            pre = 'anchor' if rows == 36 else 'cell'
            xy = np.arange(rows) % 4 < 2
            std[xy], bias[xy] = _EFDET_TARGETS[pre + '_xy']
            std[~xy], bias[~xy] = _EFDET_TARGETS[pre + '_wh']
    return std, bias


def _efdet_last(key, shape, kind, calib):
    """Final head layers: weights = unit-gain normal x (target std / spread measured under unit gain), bias =
    target - gain x (the output channel's measured mean) so every channel has the target distribution whatever
    the tower statistics are (both measurements are part of the calibration file, oracle/calibrate_bn.py).
    RetinaNet class layers (9 x 80 rows) add the anchor's objectness direction to every class row (see _EFDET_TARGETS)."""
    if '.depthwise.' in key:                       # SeparableConv2d last layer: the gain sits on the pointwise conv
        return _normal(key, shape, std=1.0 / 3.0)
    module = key.rsplit('.', 2)[0] if '.pointwise.' in key else key.rsplit('.', 1)[0]
    rows = shape[0]
    # pyramid level l has stride 8 << l in every configuration of the family (three-level nets stop at 32): counted from the
    # stride-128 level so that the shift follows the cell size, not the list position
    std, bias = _efdet_row_targets(kind, rows, max(0, 4 - int(key.split('.')[2])))
    measured = calib.get('__std__/' + module)
    gain = std / np.float32(measured) if measured is not None else np.ones(rows, np.float32)
    retina = kind == 'class' and rows == 9 * _N_CLS
    g_obj = np.float32(_EFDET_TARGETS['object'][0]) / (np.float32(measured) if measured is not None else np.float32(1.0))
    if key.endswith('.bias'):
        mean = calib.get('__mean__/' + module)
        mean = mean.astype(np.float32) if mean is not None else np.zeros(rows, np.float32)
        shift = gain * mean
        if retina:                                 # the objectness direction's own offset: its rows are averages of the unit rows
            shift = shift + g_obj * np.repeat(mean.reshape(9, _N_CLS).sum(1) / np.float32(np.sqrt(_N_CLS)), _N_CLS)
        return (bias - shift + _normal(key, shape, std=0.05)).astype(np.float32)
    fan_in = shape[1] * shape[2] * shape[3]
    unit = _normal(key, shape, std=1.0 / np.sqrt(fan_in))
    w = unit * gain.reshape(-1, 1, 1, 1)
    if retina:
        obj = unit.reshape(9, _N_CLS, -1).sum(1, keepdims=True) / np.float32(np.sqrt(_N_CLS))       # [9, 1, fan_in]
        w = w + (g_obj * np.broadcast_to(obj, (9, _N_CLS, obj.shape[-1]))).reshape(shape).astype(np.float32)
    return w.astype(np.float32)


def _efdet_bn(key, shape, damped, stiff=False):
    """BatchNorm affine parameters of the EfficientDet family by the layer's role (running statistics come from
    the calibration file)."""
    pg, pb, rg = ((_STIFF['pre_swish_gamma'], _STIFF['pre_swish_beta'], _STIFF['residual_gamma']) if stiff
                  else (_PRE_SWISH_GAMMA, _PRE_SWISH_BETA, _RESIDUAL_GAMMA))
    pre_swish = ('._bn0.' in key or '._bn1.' in key                         # stem, expand, depthwise BNs
                 or key.startswith(('fpn.', 'backbone.c5_to_c6.', 'backbone.c6_to_c7.'))   # feed fusion -> swish
                 or key.startswith(('rpn.class_nets.', 'rpn.bbox_nets.', 'rpn.center_nets.')))
    if key.endswith('.weight'):
        if key in damped:
            return _uniform(key, shape, *rg)
        return _uniform(key, shape, *pg) if pre_swish else _uniform(key, shape, 0.5, 1.5)
    return _normal(key, shape, std=pb[1], mean=pb[0]) if pre_swish else _normal(key, shape, std=0.1)


def is_efficientdet_key(key):
    return key.startswith(('backbone.model.', 'backbone.c5_to_c6.', 'backbone.c6_to_c7.', 'rpn.class_nets.',
                           'rpn.bbox_nets.', 'rpn.bbox_lasts.', 'rpn.center_nets.')) or (
        key.startswith('fpn.') and key.split('.')[1].isdigit())


def residual_project_bns(template):
    """Keys of the project-conv BatchNorm weights of the MBConv blocks that have a skip connection (stride 1 and
    Cin == Cout, external/efficientnet/model.py:94; in B0..B7 a stride-2 block always changes the width)."""
    out = set()
    for k, v in template.items():
        if k.endswith('._project_conv.weight'):
            p = k[:-len('._project_conv.weight')]
            e = template.get(p + '._expand_conv.weight')
            cin = e.shape[1] if e is not None else template[p + '._depthwise_conv.weight'].shape[0]
            if cin == v.shape[0]:
                out.add(p + '._bn2.weight')
    return out


_CALIB_CACHE = {}


def load_calibration(config_name, recipe='conditioned'):
    """BN running statistics measured once on synthetic images (oracle/calibrate_bn.py), or {}.
    Random running stats do not normalise anything, and through 23 MBConv blocks + 4 BiFPN layers the
    activations run away; the EfficientDet-family configs therefore ship calibrated statistics."""
    import os
    ck = config_name if recipe == 'conditioned' else f'{config_name}.{recipe}'
    if ck not in _CALIB_CACHE:
        stem = {'d1_fcs2': 'd1_fcs2_atss'}.get(config_name, config_name)      # same network, same statistics
        if recipe != 'conditioned':
            stem += '.' + recipe
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'calib', f'{stem}.npz')
        _CALIB_CACHE[ck] = dict(np.load(path)) if os.path.exists(path) else {}
    return _CALIB_CACHE[ck]


def is_yolov3_80(template):
    """True for the state_dict of configs/yolov3_80.json: Darknet-53 (29-entry netlist) + YOLOv3 pyramid + 255-row heads."""
    w = template.get('rpn.heads.conv_0.weight')
    return ('backbone.netlist.28.cbl_1.conv.weight' in template and 'fpn.branch_P3.cbl_0.conv.weight' in template
            and w is not None and tuple(w.shape) == (255, 256, 1, 1))


def is_ultralytics(template):
    """True for the state_dict of a model on the Ultralytics trunk (its first module is Focus: netlist.0.conv.conv)."""
    return 'backbone.netlist.0.conv.conv.weight' in template


def ultralytics_residual_bns(template):
    """BatchNorm weight keys that close a residual branch of the Ultralytics trunk: cv2 of every Bottleneck with a
    shortcut (external/ultralytics/common.py:28-37) -- the stride-4 stage `netlist.2.{i}` and the `.m.{i}` chains of the
    backbone's BottleneckCSP blocks (the pyramid's are built with shortcut=False, models/fpns.py:90-98)."""
    out = set()
    for k in template:
        if k.startswith('backbone.netlist.') and k.endswith('.cv2.bn.weight') and (k.split('.')[3] == 'm' or k.split('.')[3].isdigit()):
            out.add(k)
    return out


def make_tensor(key: str, shape, dtype=torch.float32, calib=None, damped=(), head_rms=None, stiff=False) -> torch.Tensor:
    """The synthetic value of parameter/buffer `key`.  `calib`: the calibration dict of the configuration (final-layer
    gains of the EfficientDet family); `damped`: BatchNorm weight keys of residual branches (residual_project_bns)."""
    shape = tuple(shape)
    if key.endswith('num_batches_tracked'):
        return torch.zeros(shape, dtype=torch.int64)
    effdet = is_efficientdet_key(key)
    if key.startswith('rpn.heads.conv_'):
        arr = _yolo_head(key, shape, feature_rms=head_rms, calib=calib)
    elif effdet and _efdet_last_kind(key) is not None:
        arr = _efdet_last(key, shape, _efdet_last_kind(key), calib or {})
    elif effdet and len(shape) == 1 and key.endswith(('.weight', '.bias')) and _is_bn_key(key):
        arr = _efdet_bn(key, shape, damped, stiff)
    elif key.endswith('.weights'):                           # BiFPN fusion weights (models/fpns.py:425)
        arr = _uniform(key, shape, 0.5, 1.5)
    elif key.endswith('running_var'):
        arr = _uniform(key, shape, 0.5, 1.5)
    elif key.endswith('running_mean'):
        arr = _normal(key, shape, std=0.1)
    elif len(shape) == 4:                                   # conv weight, OIHW
        fan_in = shape[1] * shape[2] * shape[3]
        arr = _normal(key, shape, std=np.sqrt(1.664 / fan_in))
    elif len(shape) == 1 and key.endswith('.weight'):       # BN gamma
        damp = ('.cbl_1.bn.' in key and key.startswith('backbone.')) or key in damped
        arr = _uniform(key, shape, 0.1, 0.3) if damp else _uniform(key, shape, 0.5, 1.5)
    elif len(shape) == 1 and key.endswith('.bias'):         # BN beta / conv bias
        arr = _normal(key, shape, std=0.1)
    else:
        arr = _normal(key, shape, std=0.1)
    return torch.from_numpy(np.ascontiguousarray(arr)).to(dtype)


def _is_bn_key(key):
    """EfficientDet-family 1-D '.weight'/'.bias' keys that belong to a BatchNorm2d (not a conv bias): _bn{0,1,2},
    the '.1' member of a conv+BN pair (c5_to_c6, p*in_*, spconv_bn, tower entries)."""
    parts = key.split('.')
    mod = parts[-2]
    if mod.startswith('_bn'):
        return True
    if mod == '1' and (key.startswith(('backbone.c5_to_c6.', 'backbone.c6_to_c7.', 'fpn.'))
                       or (key.startswith(('rpn.class_nets.', 'rpn.bbox_nets.', 'rpn.center_nets.')) and len(parts) == 6)):
        return True
    return False


def make_state_dict(template, config_name=None, recipe='conditioned') -> dict:
    """template: mapping key -> tensor (only shape/dtype are used).  With `config_name`, BN running
    statistics come from mydetection_amd/calib/<config_name>.npz when that file exists.
    recipe: 'conditioned' (default; every fixture and benchmark) or 'stiff' (EfficientDet family only: the
    ill-conditioned set described at _STIFF)."""
    assert recipe in ('conditioned', 'stiff')
    if config_name is None and is_yolov3_80(template):
        config_name = 'yolov3_80'                 # its head calibration is part of the recipe, named or not
    calib = load_calibration(config_name, recipe) if config_name else {}
    damped = residual_project_bns(template)
    head_rms = None
    if is_ultralytics(template):
        damped = damped | ultralytics_residual_bns(template)
        head_rms = _U5M_HEAD_FEATURE_RMS
    out = {}
    for k, v in template.items():
        if k in calib:
            out[k] = torch.from_numpy(np.ascontiguousarray(calib[k])).to(v.dtype)
        else:
            out[k] = make_tensor(k, v.shape, v.dtype, calib, damped, head_rms, recipe == 'stiff')
    return out


def make_images(batch: int, size, seed: int = 0, kind: str = 'rects') -> torch.Tensor:
    """Synthetic 'RGB_1' images in [0,1], NCHW float32 (utils/image_ops.py:165-188 identity case).

    kind='uniform': iid U[0,1) pixels.  kind='rects' (default): random
    coloured rectangles over a flat background plus 10% noise -- spatial structure, so head
    logits vary over the grid and the NMS workload is not one class everywhere.
    """
    h, w = (size, size) if isinstance(size, int) else size
    rng = np.random.Generator(np.random.PCG64(seed))
    if kind == 'uniform':
        return torch.from_numpy(rng.random((batch, 3, h, w), dtype=np.float32))
    out = np.empty((batch, 3, h, w), dtype=np.float32)
    for b in range(batch):
        img = np.broadcast_to(rng.random((3, 1, 1), dtype=np.float32), (3, h, w)).copy()
        for _ in range(48):
            rh = int(rng.integers(8, max(9, h // 2)))
            rw = int(rng.integers(8, max(9, w // 2)))
            y0 = int(rng.integers(0, h - 4))
            x0 = int(rng.integers(0, w - 4))
            img[:, y0:y0 + rh, x0:x0 + rw] = rng.random((3, 1, 1), dtype=np.float32)
        img += (rng.random((3, h, w), dtype=np.float32) - np.float32(0.5)) * np.float32(0.2)
        out[b] = np.clip(img, 0.0, 1.0)
    return torch.from_numpy(out)


IMAGENET_MEAN = (0.485, 0.456, 0.406)       # utils/image_ops.py:177-180 ('RGB_1_norm')
IMAGENET_STD = (0.229, 0.224, 0.225)


def make_normalized_images(batch: int, size, seed: int = 0, kind: str = 'rects') -> torch.Tensor:
    """'RGB_1_norm' inputs (EfficientDet / FCOS configs): make_images normalised by the ImageNet mean/std."""
    x = make_images(batch, size, seed, kind)
    mean = torch.tensor(IMAGENET_MEAN).view(1, 3, 1, 1)
    std = torch.tensor(IMAGENET_STD).view(1, 3, 1, 1)
    return (x - mean) / std


def make_image_set(lo: int, hi: int, size, input_format: str = 'RGB_1', base_seed: int = 1000) -> torch.Tensor:
    """Images [lo, hi) of ONE global synthetic batch: image i is a pure function of i (its own seed), so any
    contiguous sharding of the batch over ranks reproduces the same images (bench.py, multi-GPU verification).
    input_format: 'RGB_1' or 'RGB_1_norm' (general.input_format of the configuration)."""
    make = make_images if input_format == 'RGB_1' else make_normalized_images
    return torch.cat([make(1, size, seed=base_seed + i) for i in range(lo, hi)], dim=0)
