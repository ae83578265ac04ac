"""Feature pyramids of the hot path (reference: models/fpns.py)."""
import torch.nn as nn

from .. import ops
from .modules import ConvBnLeaky


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
            x = ops.upsample_concat(pre, tuple(x.shape[2:4]), x)      # cat((pre, x), dim=1)
        x = self.cbl_0(x)
        x = self.cbl_1(x)
        x = self.cbl_2(x)
        x = self.cbl_3(x)
        feature = self.cbl_4(x)
        x = self.cbl_5(feature)
        return x, feature
