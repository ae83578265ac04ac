"""Detection heads of the hot path (reference: models/rpns.py)."""
import torch.nn as nn

from .. import ops
import numpy as np
import torch

from .modules import prepare_wino, prepare_b3, FusedConvMixin, SeparableConv2d, SpconvBn, prepare_conv


class RawPreds(dict):
    """The reference's raw-prediction dict ('bbox', 'conf', 'class' views) plus a handle on the
    pixel-major head tensor the views alias, so the decode kernel can read it in place."""
    packed = None        # dict(box=(tensor, ld, astride, c0), cls=(tensor, ld, astride, c0, conf_c0)); tensors are
                         # logical [B,ch,H,W] stored channels-last with pixel stride ld


class _HeadConv(nn.Conv2d, FusedConvMixin):
    pass


class YOLOHead(nn.Module):
    '''
    One 1x1 conv (+bias) per level -> A*(bbox_param+1+n_cls) channels
    (reference: models/rpns.py:8-45).  Output channel a*(5+C)+c; the returned dict holds the
    same permuted views as the reference: 'bbox' [B,A,H,W,4], 'conf' [B,A,H,W,1], 'class' [B,A,H,W,C].
    The conv writes pixel-major rows padded to a multiple of 4 floats (255 -> 256), i.e. one
    pixel = one 1 KiB line for the decode kernel.
    '''
    def __init__(self, cfg: dict):
        super().__init__()
        self.n_anch = cfg['model.yolo.num_anchor_per_level']
        self.n_cls = cfg['general.num_class']
        self.bb_param = cfg.get('general.bbox_param', 4)
        self.heads = nn.ModuleList()
        out_ch = (self.bb_param + 1 + self.n_cls) * self.n_anch
        for i, ch in enumerate(cfg['model.fpn.out_channels']):
            self.heads.add_module(name=f'conv_{i}', module=_HeadConv(ch, out_ch, 1, stride=1, padding=0))

    def forward(self, features):
        nBp = self.bb_param
        all_level_preds = []
        for module, P in zip(self.heads, features):
            w, scale, shift = module._prepared(module, None)
            preds = ops.conv2d(P, w, scale, shift, 1, 1, (0, 0, 0, 0), ops.ACT_NONE, b3=prepare_b3(module, 'b3', w))
            nB, _, nH, nW = preds.shape
            per = nBp + 1 + self.n_cls
            raw = RawPreds()
            if self.n_anch > 1:
                v = preds.view(nB, self.n_anch, per, nH, nW)
                raw['bbox'] = v[:, :, 0:nBp, :, :].permute(0, 1, 3, 4, 2)
                raw['conf'] = v[:, :, nBp:nBp + 1, :, :].permute(0, 1, 3, 4, 2)
                raw['class'] = v[:, :, nBp + 1:, :, :].permute(0, 1, 3, 4, 2)
            else:
                raw['bbox'] = preds[:, 0:nBp, :, :].permute(0, 2, 3, 1)
                raw['conf'] = preds[:, nBp:nBp + 1, :, :].permute(0, 2, 3, 1)
                raw['class'] = preds[:, nBp + 1:, :, :].permute(0, 2, 3, 1)
            ld = ops.nhwc_ld(preds)
            raw.packed = {'box': (preds, ld, per, 0), 'cls': (preds, ld, per, nBp + 1, nBp)}
            all_level_preds.append(raw)
        return all_level_preds


def spconv3x3_bn_swish(inout_ch):
    return SpconvBn(inout_ch, swish=True)


class _LastConv(nn.Conv2d):
    """Dense 3x3 last conv of a head branch (reference: models/rpns.py:155-158, 245-266) as one launch; `out` lets
    several branches write channel ranges of one pixel-major tensor.  Without `out`, an output-channel count that is
    not a multiple of 4 (81 = conf + 80 classes) is computed as the next multiple with zero weight rows -- the extra
    channels are the padding the pixel-major tensor has anyway (ld 84) -- so that the Winograd kernels apply
    (88 -> 81 @80^2 at batch 32: 0.42 ms on the direct kernel); the caller gets the 81-channel view."""
    def forward(self, x, out=None):
        w, scale, shift = prepare_conv(self, 'main', self, None)
        cout = w.shape[0]
        if out is not None or cout % 4 == 0:
            u, u4 = prepare_wino(self, 'wino', w) if out is None and cout >= 32 else (None, None)
            return ops.conv2d(x, w, scale, shift, 3, 1, (1, 1, 1, 1), ops.ACT_NONE, out=out, wino=u, wino4=u4)
        cache = self.__dict__.setdefault('_prep_cache', {})
        hit = cache.get('padded')
        if hit is None or hit[0] is not w:
            cp = (cout + 3) // 4 * 4
            wp = w.new_zeros((cp,) + tuple(w.shape[1:]))
            wp[:cout] = w
            sp = shift.new_zeros(cp)
            sp[:cout] = shift
            sc = None
            if scale is not None:
                sc = scale.new_ones(cp)
                sc[:cout] = scale
            hit = (w, (wp, sc, sp))
            cache['padded'] = hit
        wp, sc, sp = hit[1]
        u, u4 = prepare_wino(self, 'wino_padded', wp)
        y = ops.conv2d(x, wp, sc, sp, 3, 1, (1, 1, 1, 1), ops.ACT_NONE, wino=u, wino4=u4)
        return y[:, :cout]


class EfDetHead(nn.Module):
    '''
    Per-level class and box towers: repeat x (sepconv -> BN -> swish) then a last sepconv (or dense conv)
    (reference: models/rpns.py:121-197).  Weights are not shared across levels.  Output dict per level:
    'bbox' [B,A,H,W,4] (or [B,H,W,4] when A == 1), 'class' [...,n_cls], and 'conf' = class channel 0
    when enable_conf.
    '''
    def __init__(self, cfg: dict):
        super().__init__()
        n_cls = cfg['general.num_class']
        n_anch = cfg['model.effrpn.num_anchor_per_level']
        feature_chs = cfg['model.fpn.out_channels']
        repeat = cfg['model.effrpn.repeat_num']
        bb_param = cfg.get('general.bbox_param', 4)
        enable_conf = cfg['model.effrpn.enable_conf']
        bbox_last_type = cfg.get('model.effrpn.bbox_last', 'default')
        cls_last_type = cfg.get('model.effrpn.cls_last', 'spconv')
        if bbox_last_type != 'default':
            raise NotImplementedError()
        self.class_nets = nn.ModuleList()
        self.bbox_nets = nn.ModuleList()
        cls_ch = n_anch * (1 + n_cls) if enable_conf else n_anch * n_cls
        for ch in feature_chs:
            bb_net = [spconv3x3_bn_swish(ch) for _ in range(repeat)]
            bb_net.append(SeparableConv2d(ch, n_anch * bb_param, 3, 1, padding=1))
            self.bbox_nets.append(nn.Sequential(*bb_net))
            cls_net = [spconv3x3_bn_swish(ch) for _ in range(repeat)]
            # final bias -log((1 - 0.01) / 0.01): initial confidences close to 0.01 (reference :150-158)
            if cls_last_type == 'spconv':
                cls_last = SeparableConv2d(ch, cls_ch, 3, 1, padding=1)
                cls_last.pointwise.weight.data.normal_(mean=0, std=0.1)
                cls_last.pointwise.bias.data.fill_(-np.log((1 - 0.01) / 0.01))
            elif cls_last_type == 'conv':
                cls_last = _LastConv(ch, cls_ch, 3, 1, padding=1)
                cls_last.weight.data.normal_(mean=0, std=0.1)
                cls_last.bias.data.fill_(-np.log((1 - 0.01) / 0.01))
            else:
                raise NotImplementedError()
            cls_net.append(cls_last)
            self.class_nets.append(nn.Sequential(*cls_net))
        self.n_anch = n_anch
        self.n_cls = n_cls
        self.bb_param = bb_param
        self.enable_conf = enable_conf

    def _tower_layers(self, features):
        """The repeat x (sepconv -> BN -> swish) layers of both towers of every level: layers of equal depth share one
        launch.  Returns (class features, box features) per level."""
        n = len(features)
        cls_t, box_t = list(features), list(features)
        depth = len(self.class_nets[0]) - 1
        for r in range(depth):
            outs = ops.sepconv_nodes([self.class_nets[i][r].node([cls_t[i]]) for i in range(n)]
                                     + [self.bbox_nets[i][r].node([box_t[i]]) for i in range(n)])
            cls_t, box_t = outs[:n], outs[n:]
        return cls_t, box_t

    def can_decode_retina(self, det_layers):
        """True when `decode_retina` covers this head with these decode layers."""
        from .detlayers.retinanet import RetinaLayer
        mods = [m for net in list(self.class_nets) + list(self.bbox_nets) for m in net]
        return (ops.FUSED_DECODE and not self.enable_conf and self.bb_param == 4 and 2 * len(self.class_nets) <= ops._lib.SEPCONV_MAX_NODES
                and len(det_layers) == len(self.class_nets) and all(m.fusable() for m in mods)
                and all(isinstance(net[-1], SeparableConv2d) for net in list(self.class_nets) + list(self.bbox_nets))
                and all(type(d) is RetinaLayer and d.num_anchors == self.n_anch and d.n_cls == self.n_cls for d in det_layers)
                and 65 <= self.n_cls <= 96 and self.n_anch <= 12)

    def decode_retina(self, features, det_layers, img_size, bbox, class_idx, score):
        """forward() + RetinaLayer.forward of every level + the level concatenation, with the decode in the epilogue of
        the towers' last layers (ops.sepconv_decode_retina): the class logits are never written.  Candidates of level i
        go to [n_off_i, n_off_i + A*H_i*W_i) of bbox / class_idx / score."""
        cls_t, box_t = self._tower_layers(features)
        depth = len(self.class_nets[0]) - 1
        nodes, n_off = [], 0
        offs = []
        for i, f in enumerate(features):
            offs.append(n_off)
            n_off += self.n_anch * f.shape[2] * f.shape[3]
        assert n_off == bbox.shape[1]
        for i in range(len(features)):
            nd = self.class_nets[i][depth].node_per_anchor([cls_t[i]], self.n_anch, self.n_cls)
            nd.update(kind=0, stride=det_layers[i].stride, anchors_wh=None, n_off=offs[i])
            nodes.append(nd)
        for i in range(len(features)):
            nd = self.bbox_nets[i][depth].node([box_t[i]])
            nd.update(kind=1, stride=det_layers[i].stride, anchors_wh=det_layers[i].anchor_wh.numpy(), n_off=offs[i])
            nodes.append(nd)
        ops.sepconv_decode_retina(nodes, self.n_anch, self.n_cls, img_size, bbox, class_idx, score)

    def _towers(self, features):
        """Class and box predictions of every level.  When the fused node kernel covers the tower layers, the layers of
        equal depth of all levels and both towers share ONE launch (10 nodes), so the 5x5 ... 80x80 maps of a depth
        are one grid instead of twenty launches."""
        n = len(features)
        if not all(m.fusable() for net in list(self.class_nets) + list(self.bbox_nets) for m in list(net)[:-1]):
            return [(self.class_nets[i](x), self.bbox_nets[i](x)) for i, x in enumerate(features)]
        cls_t, box_t = list(features), list(features)
        depth = len(self.class_nets[0]) - 1
        for r in range(depth):
            outs = ops.sepconv_nodes([self.class_nets[i][r].node([cls_t[i]]) for i in range(n)]
                                     + [self.bbox_nets[i][r].node([box_t[i]]) for i in range(n)])
            cls_t, box_t = outs[:n], outs[n:]
        last = [self.bbox_nets[i][depth] for i in range(n)]
        cls_last = [self.class_nets[i][depth] for i in range(n)]
        if all(isinstance(m, SeparableConv2d) and m.fusable() for m in last + cls_last):
            outs = ops.sepconv_nodes([m.node([t]) for m, t in zip(cls_last, cls_t)] + [m.node([t]) for m, t in zip(last, box_t)])
            return list(zip(outs[:n], outs[n:]))
        if all(isinstance(m, SeparableConv2d) and m.fusable() for m in last):
            box = ops.sepconv_nodes([m.node([t]) for m, t in zip(last, box_t)])
        else:
            box = [m(t) for m, t in zip(last, box_t)]
        return [(m(t), b) for m, t, b in zip(cls_last, cls_t, box)]

    def forward(self, features: list):
        all_level_preds = []
        for i, (cls_pred, bbox_pred) in enumerate(self._towers(features)):
            nB, _, nH, nW = bbox_pred.shape
            nA = self.n_anch
            per_cls = self.n_cls + 1 if self.enable_conf else self.n_cls
            packed = {'box': (bbox_pred, ops.nhwc_ld(bbox_pred), self.bb_param, 0),
                      'cls': (cls_pred, ops.nhwc_ld(cls_pred), per_cls, 1 if self.enable_conf else 0, 0)}
            if nA >= 2:
                bbox_v = bbox_pred.view(nB, nA, -1, nH, nW).permute(0, 1, 3, 4, 2)
                cls_v = cls_pred.view(nB, nA, -1, nH, nW).permute(0, 1, 3, 4, 2)
            elif nA == 1:
                assert bbox_pred.shape[1] == 4
                bbox_v = bbox_pred.permute(0, 2, 3, 1)
                cls_v = cls_pred.permute(0, 2, 3, 1)
            else:
                raise Exception()
            raw = RawPreds()
            raw['bbox'] = bbox_v
            if self.enable_conf:
                assert cls_v.shape[-1] == self.n_cls + 1
                raw['conf'] = cls_v[..., 0:1]
                raw['class'] = cls_v[..., 1:]
            else:
                assert cls_v.shape[-1] == self.n_cls
                raw['class'] = cls_v
            raw.packed = packed
            all_level_preds.append(raw)
        return all_level_preds


class EfDetHead_wCenter(nn.Module):
    '''
    EfficientDet head for FCOS with its own centerness branch (reference: models/rpns.py:232-312): per level a
    class tower (repeat x sepconv-BN-swish, dense 3x3 to n_cls [+1]), a box tower (repeat x sepconv-BN-swish) feeding
    a dense 3x3 to 4 box logits and, through one more sepconv-BN-swish, a dense 3x3 to the centerness logit.
    The class and centerness convs write channel ranges of ONE pixel-major tensor, which the decode kernel reads
    in place.  Output dict per level: 'bbox' [B,H,W,4], 'center' [B,H,W,1], 'class' [B,H,W,n_cls] (+ 'conf').
    '''
    def __init__(self, cfg: dict):
        super().__init__()
        n_cls = cfg['general.num_class']
        n_anch = cfg['model.effrpn.num_anchor_per_level']
        feature_chs = cfg['model.fpn.out_channels']
        repeat = cfg['model.effrpn.repeat_num']
        bb_param = cfg.get('general.bbox_param', 4)
        enable_conf = cfg.get('model.effrpn.enable_conf', False)
        assert n_anch == 1
        assert cfg['model.effrpn.enable_centerscore']
        self.class_nets = nn.ModuleList()
        self.bbox_nets = nn.ModuleList()
        self.bbox_lasts = nn.ModuleList()
        self.center_nets = nn.ModuleList()
        for ch in feature_chs:
            self.bbox_nets.append(nn.Sequential(*[spconv3x3_bn_swish(ch) for _ in range(repeat)]))
            self.bbox_lasts.append(_LastConv(ch, n_anch * bb_param, 3, 1, padding=1))
            # final biases -log((1 - 0.05) / 0.05): initial confidences close to 0.05 (reference :250-264)
            ct_last = _LastConv(ch, 1, kernel_size=3, stride=1, padding=1)
            ct_last.weight.data.normal_(mean=0, std=0.01)
            ct_last.bias.data.fill_(-np.log((1 - 0.05) / 0.05))
            self.center_nets.append(nn.Sequential(spconv3x3_bn_swish(ch), ct_last))
            cls_net = [spconv3x3_bn_swish(ch) for _ in range(repeat)]
            cls_last = _LastConv(ch, n_anch * (1 + n_cls) if enable_conf else n_anch * n_cls, 3, 1, padding=1)
            cls_last.weight.data.normal_(mean=0, std=0.01)
            cls_last.bias.data.fill_(-np.log((1 - 0.05) / 0.05))
            cls_net.append(cls_last)
            self.class_nets.append(nn.Sequential(*cls_net))
        self.n_cls = n_cls
        self.enable_conf = enable_conf

    def forward(self, features: list):
        all_level_preds = []
        cls_ch = self.n_cls + 1 if self.enable_conf else self.n_cls
        for i, x in enumerate(features):
            nB, _, nH, nW = x.shape
            # [B,H,W,ld]: channels [0, cls_ch) class (+conf) logits, channel cls_ch the centerness logit
            both, ld = ops.empty_nhwc(nB, cls_ch + 1, nH, nW, x.device)
            t = x
            for m in list(self.class_nets[i])[:-1]:
                t = m(t)
            self.class_nets[i][-1](t, out=both[:, :cls_ch])
            bbox_feats = self.bbox_nets[i](x)
            bbox_pred = self.bbox_lasts[i](bbox_feats)
            self.center_nets[i][1](self.center_nets[i][0](bbox_feats), out=both[:, cls_ch:cls_ch + 1])
            assert bbox_pred.shape[1] == 4
            cls_v = both.permute(0, 2, 3, 1)
            raw = RawPreds()
            raw['bbox'] = bbox_pred.permute(0, 2, 3, 1)
            raw['center'] = cls_v[..., cls_ch:cls_ch + 1]
            if self.enable_conf:
                raw['conf'] = cls_v[..., 0:1]
                raw['class'] = cls_v[..., 1:cls_ch]
            else:
                raw['class'] = cls_v[..., :cls_ch]
            raw.packed = {'box': (bbox_pred, ops.nhwc_ld(bbox_pred), 4, 0),
                          'cls': (both, ld, cls_ch + 1, 1 if self.enable_conf else 0, cls_ch)}
            all_level_preds.append(raw)
        return all_level_preds
