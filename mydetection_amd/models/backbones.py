"""Backbones of the hot path (reference: models/backbones.py)."""
import torch.nn as nn

from .. import ops
from .modules import ConvBn, ConvBnLeaky, DarkBlock


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


class UltralyticsBackbone(nn.Module):
    """YOLOv5 trunk (reference: models/backbones.py:60-113): Focus, then four stride-2 Conv stages with Bottleneck /
    BottleneckCSP / SPP blocks scaled by model.ultralytics.depth_muliple / channel_muliple (the reference's spelling);
    returns the features at strides 8, 16, 32.  netlist indices, and with them the state_dict keys, are the reference's."""
    def __init__(self, global_cfg):
        super().__init__()
        import math
        from ..external.ultralytics.common import Focus, Bottleneck, BottleneckCSP, Conv, SPP
        depm = global_cfg['model.ultralytics.depth_muliple']
        chm = global_cfg['model.ultralytics.channel_muliple']
        ch = [int(math.ceil(c * chm / 8) * 8) for c in (64, 128, 256, 512, 1024)]

        def reps(n):
            return max(round(n * depm), 1)
        self.feature_chs = ch[2:]
        self.feature_strides = [8, 16, 32]
        self.netlist = nn.ModuleList()
        if global_cfg['model.ultralytics.first'] == 'Focus':
            assert global_cfg.get('general.input.frame_concatenation', 1) == 1
            self.netlist.append(Focus(3, ch[0], k=3))
        else:                                   # 'Conv2d' raises in the reference too (models/backbones.py:82-83)
            raise NotImplementedError()
        self.netlist.append(Conv(ch[0], ch[1], k=3, s=2))                                        # 4x
        self.netlist.append(nn.Sequential(*[Bottleneck(ch[1], ch[1]) for _ in range(reps(3))]))
        self.netlist.append(Conv(ch[1], ch[2], k=3, s=2))                                        # 8x
        self.netlist.append(BottleneckCSP(ch[2], ch[2], n=reps(9)))
        self.netlist.append(Conv(ch[2], ch[3], k=3, s=2))                                        # 16x
        self.netlist.append(BottleneckCSP(ch[3], ch[3], n=reps(9)))
        self.netlist.append(Conv(ch[3], ch[4], k=3, s=2))                                        # 32x
        self.netlist.append(SPP(ch[4], ch[4], k=[5, 9, 13]))
        self.netlist.append(BottleneckCSP(ch[4], ch[4], n=reps(6)))

    def forward(self, x):
        assert x.dim() == 4 and x.shape[2] % 32 == 0 and x.shape[3] % 32 == 0
        features = []
        for module in self.netlist:             # a feature is tapped wherever the next module halves the map
            y = module(x)
            if y.shape[2:4] != x.shape[2:4]:
                assert y.shape[2] == x.shape[2] // 2
                features.append(x)
            x = y
        features.append(x)
        assert len(features) == 6
        return features[3:]


class _PoolAfter(ConvBn):
    """nn.Sequential(Conv2d, BatchNorm2d, MaxPool2d(3, 2, 1)): fused conv+BN launch, then the pool kernel."""
    def __init__(self, in_ch, out_ch, k, padding):
        super().__init__(in_ch, out_ch, k, padding)
        self.add_module('2', nn.MaxPool2d(3, stride=2, padding=1))       # parameter-free; keeps the index layout

    def forward(self, x):
        return ops.maxpool3s2(ConvBn.forward(self, x))


class _MaxPool(nn.Module):
    def forward(self, x):
        return ops.maxpool3s2(x)


class EfNetBackbone(nn.Module):
    '''
    EfficientNet feature extractor + C6/C7 (reference: models/backbones.py:155-232).  Returns
    [C3, C4, C5] or [C3, C4, C5, C6, C7]; a feature is tapped whenever the spatial size changes.
    '''
    valid_names = {'efficientnet-b0', 'efficientnet-b1', 'efficientnet-b2', 'efficientnet-b3', 'efficientnet-b4',
                   'efficientnet-b5', 'efficientnet-b6'}

    def __init__(self, cfg: dict):
        super().__init__()
        model_name = cfg['model.backbone.name']
        assert model_name in self.valid_names, 'Unknown efficientnet model name'
        from ..external.efficientnet.model import EfficientNet
        efn = EfficientNet.from_name(model_name)
        self.model = efn
        efnet_chs = [efn._blocks_args[i].output_filters for i in [2, 4, 6]]
        if cfg['model.backbone.num_levels'] == 3:
            self.feature_chs = efnet_chs
            self.feature_strides = (8, 16, 32)
            self.C6C7 = False
        elif cfg['model.backbone.num_levels'] == 5:
            out_ch = cfg['model.backbone.C6C7_out_channels']
            downsample_layer = cfg.get('model.efficientnet.C6C7_downsample', 'maxpool')
            if downsample_layer == 'maxpool':
                self.c5_to_c6 = _PoolAfter(efnet_chs[-1], out_ch, 1, 0)
                self.c6_to_c7 = _MaxPool()
            elif downsample_layer == 'conv':
                self.c5_to_c6 = _PoolAfter(efnet_chs[-1], out_ch, 3, 1)
                self.c6_to_c7 = _PoolAfter(out_ch, out_ch, 3, 1)
            else:
                raise NotImplementedError()
            self.feature_chs = efnet_chs + [out_ch, out_ch]
            self.feature_strides = (8, 16, 32, 64, 128)
            self.C6C7 = True
        else:
            raise NotImplementedError()
        self.enable_dropout = cfg['model.efficientnet.enable_dropout']     # drop-connect: identity at inference

    def forward(self, x):
        blocks = list(self.model._blocks)
        if self.model.stem_fusable():           # stem + block 0's depthwise conv in one launch (block 0 keeps the resolution)
            x = self.model.stem_block0(x)
            blocks = blocks[1:]
        else:
            x = self.model.stem(x)
        features = []
        for block in blocks:
            y = block(x)
            if y.shape[-1] != x.shape[-1]:
                features.append(x)
            x = y
        features.append(x)
        C1, C2, C3, C4, C5 = features
        if self.C6C7:
            C6 = self.c5_to_c6(C5)
            C7 = self.c6_to_c7(C6)
            return [C3, C4, C5, C6, C7]
        return [C3, C4, C5]
