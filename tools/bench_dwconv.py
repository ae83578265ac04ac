"""Time the depthwise kernel (mydet_dwconv_f32, squeeze-fused form) on the EfficientNet-B1 shapes at 640x640.
    python tools/bench_dwconv.py [--batch 16]        MYDET_DW_TILED=0 selects the register-blocked kernels for stride 1"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mydetection_amd import ops                                           # noqa: E402
from mydetection_amd.external.efficientnet.model import static_same_pad  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=16)
a = ap.parse_args()
dev = torch.device('cuda')
B = a.batch
g = torch.Generator().manual_seed(0)
# (k, stride, C, hw): external/efficientnet/utils.py:258-263 scaled to B1
SHAPES = [(3, 1, 32, 320), (3, 1, 16, 320), (5, 1, 240, 80), (3, 2, 240, 80), (3, 1, 480, 40), (5, 1, 480, 40), (5, 1, 672, 40),
          (5, 2, 672, 40), (5, 1, 1152, 20), (3, 1, 1152, 20), (3, 1, 1920, 20)]


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


for k, s, c, hw in SHAPES:
    x = torch.randn(B, hw, hw, c, generator=g).to(dev).permute(0, 3, 1, 2)
    w = (torch.randn(k, k, c, generator=g) / k).to(dev)
    sc, sh = (torch.rand(c, generator=g) + 0.5).to(dev), torch.randn(c, generator=g).to(dev)
    pad = static_same_pad(k, s, 240)
    ho = (hw + pad[0] + pad[2] - k) // s + 1
    us = timeit(lambda: ops.dwconv(x, w, sc, sh, k, s, pad, ops.ACT_SWISH, squeeze=True))
    nbytes = B * c * 4 * (hw * hw + ho * ho)
    print(f'dw k{k}s{s} {c:5d} ch {hw:3d}x{hw:<3d} {us:8.1f} us  {nbytes / us / 1e3:7.1f} GB/s (in + out once)')
