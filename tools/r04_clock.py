"""In-kernel clock and K-loop cycles of the F(4x4) GEMM launch, per workgroup.  Needs the DIAGNOSTIC build of the library:
    make -C mydetection_amd/csrc clean all EXTRA=-DMYDET_DIAG        (rebuild without EXTRA afterwards: its results are not valid)
    MYDET_W4_DBG=8 python tools/r04_clock.py <Cin> <Cout> <H=W> <batch>      8 = stamps; + 1 every stage the same 36 KB,
                                                                            + 2 no DMA after stage 1, + 3 U always stage 0
The stamped build writes (delta s_memtime, delta s_memrealtime) per workgroup into the consumed V workspace.
Round-4 results: profiles/r04_experiments/exp5.txt, exp6.txt; profiles/HISTORY.md."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mydetection_amd import ops
cin, cout, hw, B = (int(v) for v in sys.argv[1:5])
dev = torch.device('cuda')
x = torch.randn(B, hw, hw, cin, device=dev).permute(0, 3, 1, 2)
w = (torch.randn(cout, 3, 3, cin, device=dev) / (cin * 9) ** 0.5).contiguous()
sc, sh = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev) * 0.1
res = torch.randn(B, hw, hw, cout, device=dev).permute(0, 3, 1, 2)
u4 = ops.wino4_weights(w)
t_end = time.time() + float(os.environ.get('SOAK_S', '2.0'))
n = 0
while time.time() < t_end:                      # back-to-back launches so that the clock settles
    for _ in range(20):
        ops.conv2d(x, w, sc, sh, 3, 1, (1, 1, 1, 1), 1, residual=res, wino4=u4)
    torch.cuda.synchronize(); n += 20
ws = list(ops._WINO4_WS.values())[0].view(torch.int32).cpu().numpy()
nk = cin // 4
MT = B * ((hw + 3) // 4) ** 2
nmb, ntn = (MT + 31) // 32, (cout + 31) // 32
V4 = 9 * 4 * 32 * 4                                 # floats per stage of V
cyc, rt = [], []
for mb in range(nmb):
    base = mb * nk * V4
    d = ws[base: base + ntn * 4].reshape(ntn, 4)
    cyc += list(d[:, 0]); rt += list(d[:, 1])
cyc, rt = np.array(cyc, dtype=np.float64), np.array(rt, dtype=np.float64)
ok = rt > 0
ghz = cyc[ok] / rt[ok] * 0.1
ideal = nk * 36 * 32 * 2                            # cycles of the matrix pipe per SIMD shared by two workgroups
import collections
hist = collections.Counter((cyc[ok] // 50000).astype(int))
print("  loop-cycle histogram (x50k):", sorted(hist.items()))
print(f'dbg={os.environ.get("MYDET_W4_DBG")} {cin}->{cout} {hw}x{hw} b{B}: {n} launches; clock median {np.median(ghz):.3f} GHz (p10 {np.percentile(ghz, 10):.3f}, p90 {np.percentile(ghz, 90):.3f}); '
      f'main loop median {np.median(cyc[ok]):.0f} cycles = {np.median(rt[ok]) / 100:.1f} us (pipe-bound floor for two co-resident workgroups {ideal} cycles -> ratio {ideal / np.median(cyc[ok]):.3f})')
