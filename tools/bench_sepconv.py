"""Time the fused pyramid node kernel (mydet_sepconv_nodes_f32) on the D1 shapes, batch 16 at 640x640:
a head-tower launch (10 nodes), the last-layer launch (88->720 / 88->36), and single BiFPN nodes per level.
hipGraph replay of 20 launches, HIP events.   python tools/bench_sepconv.py [--batch 16]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mydetection_amd import ops                                           # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=16)
ap.add_argument('--only', default='', help='run only the cases whose name contains this')
a = ap.parse_args()
dev = torch.device('cuda')
B, C = a.batch, 88
g = torch.Generator(device='cpu').manual_seed(0)
LEVELS = [80, 40, 20, 10, 5]


def feat(hw):
    return torch.randn(B, hw, hw, C, generator=g).to(dev).permute(0, 3, 1, 2)


def node(inputs, modes, cout, act, bn=True):
    return dict(inputs=inputs, modes=modes, fuse_weights=torch.tensor([0.9, 1.1, 0.7][:len(inputs)]).to(dev) if len(inputs) > 1 else None,
                w_dw=(torch.randn(3, 3, C, generator=g) / 3).to(dev), w_pw=ops.pack_pointwise((torch.randn(cout, C, generator=g) / 9).to(dev)),
                scale=(torch.rand(cout, generator=g) + 0.5).to(dev) if bn else None, shift=torch.randn(cout, generator=g).to(dev),
                cout=cout, act=act)


def timeit(name, nodes, flops, nbytes):
    if a.only and a.only not in name:
        return
    run = lambda: ops.sepconv_nodes(nodes)                                # noqa: E731
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(3):
            run()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=side):
        for _ in range(20):
            run()
    gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        gr.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 100 * 1e3
    print(f'{name:44s} {us:8.1f} us  {flops / us / 1e6:7.1f} TFLOP/s  {nbytes / us / 1e3:7.1f} GB/s (in+out once)')


px = [B * h * h for h in LEVELS]
feats = [feat(h) for h in LEVELS]
timeit('tower depth (5 levels x 2 towers, 88->88)', [node([f], [0], 88, 2) for f in feats] * 2,
       2 * sum(px) * 2 * 88 * 88, 2 * sum(px) * 2 * 88 * 4)
timeit('last layers (5 x 88->720, 5 x 88->36)', [node([f], [0], 720, 0, False) for f in feats] + [node([f], [0], 36, 0, False) for f in feats],
       sum(px) * 2 * 88 * 756, sum(px) * (2 * 88 + 756) * 4)
for i, h in enumerate(LEVELS):
    timeit(f'tower node alone {h}x{h}', [node([feats[i]], [0], 88, 2)], px[i] * 2 * 88 * 88, px[i] * 2 * 88 * 4)
for i, h in enumerate(LEVELS[:-1]):
    timeit(f'BiFPN top-down node {h}x{h} (same + up2x)', [node([feats[i], feats[i + 1]], [0, 1], 88, 0)], px[i] * 2 * 88 * 88,
           (px[i] * 2 + px[i + 1]) * 88 * 4)
for i, h in enumerate(LEVELS[1:], start=1):
    timeit(f'BiFPN bottom-up node {h}x{h} (same + same + pool)', [node([feats[i], feats[i], feats[i - 1]], [0, 0, 2], 88, 0)],
           px[i] * 2 * 88 * 88, (px[i] * 3 + px[i - 1]) * 88 * 4)

