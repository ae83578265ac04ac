"""Time the fused MBConv front half (mydet_mbconv_expand_dw_f32) against the two launches it replaces on the shallow
EfficientNet-B1 blocks at 640x640.   python tools/bench_mbconv.py [--batch 16] [--only fused|split]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mydetection_amd import ops                                           # noqa: E402
from mydetection_amd.external.efficientnet.model import static_same_pad  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=16)
ap.add_argument('--only', default='')
a = ap.parse_args()
dev = torch.device('cuda')
B = a.batch
g = torch.Generator().manual_seed(0)
SHAPES = [(3, 2, 16, 320), (3, 1, 24, 160), (5, 2, 24, 160), (5, 1, 40, 80), (3, 2, 40, 80)]


def timeit(run):
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(3):
            run()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=side):
        for _ in range(10):
            run()
    gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        gr.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 50 * 1e3


for k, s, cin, hw in SHAPES:
    cexp = cin * 6
    x = torch.randn(B, hw, hw, cin, generator=g).to(dev).permute(0, 3, 1, 2)
    we = (torch.randn(cexp, 1, 1, cin, generator=g) / cin ** 0.5).to(dev)
    wd = (torch.randn(k, k, cexp, generator=g) / k).to(dev)
    sc0, sh0, sc1, sh1 = [(torch.rand(cexp, generator=g) + 0.5).to(dev) for _ in range(4)]
    pad = static_same_pad(k, s, 240)
    ho = (hw + pad[0] + pad[2] - k) // s + 1
    out_mb = B * ho * ho * cexp * 4 / 1e6

    wef, wdf = ops.fold_scale(we, sc0), ops.fold_scale(wd, sc1)

    def fused():
        return ops.mbconv_expand_dw(x, wef, sh0, wdf, sh1, k, s, pad)

    def split():
        e = ops.conv2d(x, we, sc0, sh0, 1, 1, (0, 0, 0, 0), ops.ACT_SWISH)
        return ops.dwconv(e, wd, sc1, sh1, k, s, pad, ops.ACT_SWISH, squeeze=True)
    tf = timeit(fused) if a.only != 'split' else float('nan')
    ts = timeit(split) if a.only != 'fused' else float('nan')
    print(f'{cin:3d}->{cexp:4d} k{k}s{s} {hw}x{hw}: fused {tf:7.1f} us   expand+dw {ts:7.1f} us   output {out_mb:6.1f} MB '
          f'({out_mb / 4.5e6 * 1e6:5.1f} us at 4.5 TB/s)')
