"""Time the squeeze-excite tail (mydet_se_gate_f32) on the EfficientNet-B1 block shapes at 640x640.
    python tools/bench_se.py [--batch 16]
hipGraph of 20 launches per shape, HIP events; prints us per launch."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mydetection_amd import ops                                           # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=16)
a = ap.parse_args()
dev = torch.device('cuda')
B = a.batch
# (C, Cse, S, HW): S = slices the producing depthwise kernel writes
SHAPES = [(32, 8, 800, 320 * 320), (16, 4, 128, 320 * 320), (96, 4, 800, 160 * 160), (144, 6, 200, 160 * 160), (144, 6, 200, 80 * 80),
          (240, 10, 50, 80 * 80), (240, 10, 50, 40 * 40), (480, 20, 15, 40 * 40), (480, 20, 15, 40 * 40), (672, 28, 15, 40 * 40),
          (672, 28, 6, 20 * 20), (1152, 48, 6, 20 * 20), (1920, 80, 6, 20 * 20)]
for C, Cse, S, HW in SHAPES:
    partial = torch.randn(B, S + 1, C, device=dev)
    w1, b1 = torch.randn(Cse, C, device=dev) / C ** 0.5, torch.randn(Cse, device=dev)
    w2t, b2 = torch.randn(Cse, C, device=dev), torch.randn(C, device=dev)
    run = lambda: ops.se_gate(partial, HW, w1, b1, w2t, b2)              # noqa: E731
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
    print(f'C {C:5d} Cse {Cse:3d} S {S:4d}   {e0.elapsed_time(e1) / 100 * 1e3:7.2f} us')
