"""Backbones of the hot path (reference: models/backbones.py)."""
import torch.nn as nn

from .modules import ConvBnLeaky, DarkBlock


class Darknet53(nn.Module):
    """Darknet-53 trunk; returns [C3, C4, C5] at strides 8/16/32 (reference: models/backbones.py:6-57).

    netlist indices and therefore state_dict keys match the reference: 0 stem, then for each
    stage a stride-2 ConvBnLeaky followed by (1, 2, 8, 8, 4) DarkBlocks; taps after 14, 23, 28.
    """
    def __init__(self, global_cfg):
        super().__init__()
        self.netlist = nn.ModuleList()
        in_ch = global_cfg.get('model.backbone.input_channels', 3)
        self.netlist.append(ConvBnLeaky(in_ch, 32, k=3, s=1))
        width = 32
        for n_blocks in (1, 2, 8, 8, 4):
            self.netlist.append(ConvBnLeaky(width, width * 2, k=3, s=2))
            width *= 2
            for _ in range(n_blocks):
                self.netlist.append(DarkBlock(in_out=width, hidden=width // 2))
        assert len(self.netlist) == 29

    def forward(self, x):
        feats = []
        for i, layer in enumerate(self.netlist):
            x = layer(x)
            if i in (14, 23, 28):
                feats.append(x)
        return feats
