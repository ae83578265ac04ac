"""In-kernel clock of the F(4x4) GEMM launch while it alternates with the DarkBlock's 1x1 conv on (a) the float32 matrix
instruction, (b) the split-bf16 kernel -- the "every other kernel runs slower next to the bf16 kernels" observation of round 5
measured where it happens.  Needs the DIAGNOSTIC build (see tools/r04_clock.py):
    make -C mydetection_amd/csrc clean all EXTRA=-DMYDET_DIAG ; MYDET_W4_DBG=8 python tools/r05_clock_mix.py
"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mydetection_amd import ops
dev = torch.device('cuda')
B, hw, c1, c2 = 32, 80, 256, 128                   # DarkBlock at 80^2: 256 -> 128 (1x1) -> 256 (3x3) + skip
x = torch.randn(B, hw, hw, c1, device=dev).permute(0, 3, 1, 2)
w1 = (torch.randn(c2, 1, 1, c1, device=dev) / c1 ** 0.5).contiguous()
w3 = (torch.randn(c1, 3, 3, c2, device=dev) / (c2 * 9) ** 0.5).contiguous()
s1, h1 = torch.rand(c2, device=dev) + 0.5, torch.randn(c2, device=dev) * 0.1
s3, h3 = torch.rand(c1, device=dev) + 0.5, torch.randn(c1, device=dev) * 0.1
u4 = ops.wino4_weights(w3)
p3 = ops.split_bf16(w1)
y_fixed = torch.randn(B, hw, hw, c2, device=dev).permute(0, 3, 1, 2)


def clocks():
    ws = list(ops._WINO4_WS.values())[0].view(torch.int32).cpu().numpy()
    nk = c2 // 4
    MT = B * ((hw + 3) // 4) ** 2
    nmb, ntn = (MT + 31) // 32, (c1 + 31) // 32
    V4 = 9 * 4 * 32 * 4
    cyc, rt = [], []
    for mb in range(nmb):
        d = ws[mb * nk * V4: mb * nk * V4 + ntn * 4].reshape(ntn, 4)
        cyc += list(d[:, 0]); rt += list(d[:, 1])
    cyc, rt = np.array(cyc, dtype=np.float64), np.array(rt, dtype=np.float64)
    ok = rt > 0
    return cyc[ok] / rt[ok] * 0.1, cyc[ok]


for rep in range(2):
    for mode in ('float32 1x1', 'split-bf16 1x1', 'no 1x1 (F(4x4) back to back)'):
        t_end = time.time() + 2.5
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 0
        e0.record()
        while time.time() < t_end:
            for _ in range(10):
                if mode.startswith('no'):
                    y1 = y_fixed
                else:
                    y1 = ops.conv2d(x, w1, s1, h1, 1, 1, (0, 0, 0, 0), 1, b3=p3 if mode.startswith('split') else None)
                ops.conv2d(y1, w3, s3, h3, 3, 1, (1, 1, 1, 1), 1, residual=x, wino4=u4)
            torch.cuda.synchronize(); n += 10
        e1.record(); e1.synchronize()
        ghz, cyc = clocks()
        print(f'{mode:32s} pairs {n:5d}  {e0.elapsed_time(e1) / n:7.4f} ms per pair | F(4x4) GEMM launch: in-kernel clock median '
              f'{np.median(ghz):.3f} GHz (p10 {np.percentile(ghz, 10):.3f}, p90 {np.percentile(ghz, 90):.3f}), K-loop cycles median {np.median(cyc):.0f}', flush=True)
