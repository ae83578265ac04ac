"""Detection heads of the hot path (reference: models/rpns.py)."""
import torch.nn as nn

from .. import ops
from .modules import FusedConvMixin


class RawPreds(dict):
    """The reference's raw-prediction dict ('bbox', 'conf', 'class' views) plus a handle on the
    pixel-major head tensor the views alias, so the decode kernel can read it in place."""
    packed = None        # (tensor [B,ch,H,W] channels-last, ld, n_anchor, per-anchor channel count)


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
            preds = ops.conv2d(P, w, scale, shift, 1, 1, (0, 0, 0, 0), ops.ACT_NONE)
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
            raw.packed = (preds, ops.nhwc_ld(preds), self.n_anch, per)
            all_level_preds.append(raw)
        return all_level_preds
