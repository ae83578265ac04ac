"""Single conv layer micro-benchmark through the C ABI (HIP events), for tuning and PMC runs.
    python tools/bench_conv.py --cin 128 --cout 256 --k 3 --s 1 --hw 80 --batch 32 [--res] [--reps 20]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mydetection_amd import ops                                           # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--cin', type=int, default=128)
ap.add_argument('--cout', type=int, default=256)
ap.add_argument('--k', type=int, default=3)
ap.add_argument('--s', type=int, default=1)
ap.add_argument('--hw', type=int, default=80)
ap.add_argument('--batch', type=int, default=32)
ap.add_argument('--res', action='store_true')
ap.add_argument('--gate', action='store_true')
ap.add_argument('--act', type=int, default=1)
ap.add_argument('--reps', type=int, default=20)
ap.add_argument('--wino', action='store_true', help='3x3 s1: fused Winograd kernel instead of the direct one')
ap.add_argument('--wino4', action='store_true', help='3x3 s1: F(4x4,3x3) Winograd kernel')
a = ap.parse_args()
dev = torch.device('cuda')
x = torch.randn(a.batch, a.hw, a.hw, a.cin, device=dev).permute(0, 3, 1, 2)
w = (torch.randn(a.cout, a.k, a.k, a.cin, device=dev) / (a.cin * a.k * a.k) ** 0.5).contiguous()
scale = torch.rand(a.cout, device=dev) + 0.5
shift = torch.randn(a.cout, device=dev) * 0.1
p = (a.k - 1) // 2
ho = (a.hw + 2 * p - a.k) // a.s + 1
res = torch.randn(a.batch, ho, ho, a.cout, device=dev).permute(0, 3, 1, 2) if a.res else None
gate = torch.rand(a.batch, a.cin, device=dev) if a.gate else None
u = ops.wino_weights(w) if a.wino else None
u4 = ops.wino4_weights(w) if a.wino4 else None
assert not a.wino or u is not None
for _ in range(3):
    y = ops.conv2d(x, w, scale, shift, a.k, a.s, (p, p, p, p), a.act, residual=res, gate=gate, wino=u, wino4=u4)
torch.cuda.synchronize()
t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0.record()
for _ in range(a.reps):
    y = ops.conv2d(x, w, scale, shift, a.k, a.s, (p, p, p, p), a.act, residual=res, gate=gate, wino=u, wino4=u4)
t1.record()
torch.cuda.synchronize()
ms = t0.elapsed_time(t1) / a.reps
fl = 2.0 * a.batch * ho * ho * a.cout * a.k * a.k * a.cin
gb = 4.0 * (a.batch * a.hw * a.hw * a.cin + a.batch * ho * ho * a.cout * (2 if a.res else 1)) / 1e9
print(f'{"wino4 " if a.wino4 else "wino " if a.wino else ""}{a.cin}->{a.cout} k{a.k}s{a.s} {a.hw}x{a.hw} b{a.batch} res={a.res}: {ms:.4f} ms  {fl / ms / 1e9:.1f} TFLOP/s  '
      f'{100 * fl / ms / 1e9 / 157.3:.1f}% of peak  {gb / ms * 1e3:.0f} GB/s')
