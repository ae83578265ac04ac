"""One-process sweep of the implicit-GEMM tile configurations (MYDET_CONV_CFG) over the pointwise layers of
EfficientNet-B1 / BiFPN / EfDetHead at 640x640.   python tools/sweep_pointwise.py [--batch 16] [--cfgs 0,1,2,3,6,8]
Prints per shape the time of the default choice and of every forced configuration (hipGraph of 10 launches)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mydetection_amd import ops                                           # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=16)
ap.add_argument('--cfgs', default='0,1,2,3,6,8')
ap.add_argument('--first', type=int, default=0, help='only the first N shapes (7 = the ones the skinny kernel of pointwise.hip takes)')
a = ap.parse_args()
dev = torch.device('cuda')
B = a.batch
# (cin, cout, hw, act, gate, res)
SHAPES = [(32, 16, 320, 0, 1, 0), (16, 16, 320, 0, 1, 1), (96, 24, 160, 0, 1, 0), (144, 24, 160, 0, 1, 1), (144, 40, 80, 0, 1, 0),
          (40, 240, 80, 2, 0, 0), (240, 40, 80, 0, 1, 1), (240, 80, 40, 0, 1, 0), (80, 480, 40, 2, 0, 0), (480, 80, 40, 0, 1, 1),
          (480, 112, 40, 0, 1, 0), (112, 672, 40, 2, 0, 0), (672, 112, 40, 0, 1, 1), (672, 192, 20, 0, 1, 0),
          (192, 1152, 20, 2, 0, 0), (1152, 192, 20, 0, 1, 1), (1152, 320, 20, 0, 1, 0), (320, 1920, 20, 2, 0, 0),
          (1920, 320, 20, 0, 1, 1), (40, 88, 80, 0, 0, 0), (112, 88, 40, 0, 0, 0), (320, 88, 20, 0, 0, 0)]


def timeit(run):
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(2):
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
    for _ in range(3):
        gr.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 30 * 1e3


cfgs = [int(c) for c in a.cfgs.split(',')]
print(f'{"shape":28s} {"default":>9s} {"pw.hip":>9s} ' + ' '.join(f'cfg{c:>2d}    ' for c in cfgs))
for cin, cout, hw, act, gate, res in (SHAPES[:a.first] if a.first else SHAPES):
    x = torch.randn(B, hw, hw, cin, device=dev).permute(0, 3, 1, 2)
    w = (torch.randn(cout, 1, 1, cin, device=dev) / cin ** 0.5).contiguous()
    sc, sh = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev) * 0.1
    r = torch.randn(B, hw, hw, cout, device=dev).permute(0, 3, 1, 2) if res else None
    g = torch.rand(B, cin, device=dev) if gate else None
    run = lambda: ops.conv2d(x, w, sc, sh, 1, 1, (0, 0, 0, 0), act, residual=r, gate=g)      # noqa: E731
    os.environ.pop('MYDET_CONV_CFG', None)
    t0 = timeit(run)
    os.environ['MYDET_PW_WIDE'] = '1'
    tw = timeit(run)
    os.environ.pop('MYDET_PW_WIDE')
    row = []
    for c in cfgs:
        os.environ['MYDET_CONV_CFG'] = str(c)
        try:
            row.append(timeit(run))
        except Exception:
            row.append(float('nan'))
    os.environ.pop('MYDET_CONV_CFG', None)
    best = min(row)
    print(f'{cin:5d}->{cout:<5d}@{hw:<3d} g{gate} r{res} a{act}   {t0:8.1f}  {tw:8.1f}  ' + ' '.join(f'{t:8.1f}{"*" if t == best else " "}' for t in row))
