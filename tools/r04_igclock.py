"""Main-loop cycles of conv_igemm_kernel per workgroup (diagnostic build: MYDET_IG_DBG = 8 + bit0 no global loads + bit1 no LDS stores)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mydetection_amd import ops
cin, cout, k, s, hw, B = (int(v) for v in sys.argv[1:7])
dev = torch.device('cuda')
x = torch.randn(B, hw, hw, cin, device=dev).permute(0, 3, 1, 2)
w = (torch.randn(cout, k, k, cin, device=dev) / (cin * k * k) ** 0.5).contiguous()
sc, sh = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev) * 0.1
p = (k - 1) // 2
ws = ops.conv_workspace(dev); ws.zero_()
t_end = time.time() + 0.7
while time.time() < t_end:
    for _ in range(20):
        ops.conv2d(x, w, sc, sh, k, s, (p, p, p, p), 1)
    torch.cuda.synchronize()
t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0.record()
for _ in range(20):
    ops.conv2d(x, w, sc, sh, k, s, (p, p, p, p), 1)
t1.record(); torch.cuda.synchronize()
ms = t0.elapsed_time(t1) / 20
d = ws.view(torch.int32).cpu().numpy()[:400000].reshape(-1, 2)
d = d[d[:, 1] > 0]
cyc, rt = d[:, 0].astype(np.float64), d[:, 1].astype(np.float64)
ho = (hw + 2 * p - k) // s + 1
fl = 2.0 * B * ho * ho * cout * k * k * cin
print(f'dbg={os.environ.get("MYDET_IG_DBG")} cfg={os.environ.get("MYDET_CONV_CFG")} {cin}->{cout} k{k}s{s} {hw}x{hw} b{B}: {ms*1e3:.1f} us = {fl/ms/1e9/157.3:.3f} of peak; '
      f'{len(cyc)} workgroups stamped, main loop median {np.median(cyc):.0f} cycles = {np.median(rt)/100:.2f} us, clock {np.median(cyc/rt)*0.1:.2f} GHz')
