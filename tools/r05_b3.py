"""Split-bf16 implicit GEMM (mydet_conv2d_igemm_b3_f32) against float64 and against the float32 kernel, with timings, on the
headline's stride-2 3x3 layers and a few 1x1 layers (batch 32 unless the map is small)."""
import sys, time
import torch
import torch.nn.functional as F
sys.path.insert(0, '.')
from mydetection_amd import ops

dev = torch.device('cuda:0')
shapes = [  # B, Cin, Cout, k, s, H
    (2, 32, 64, 3, 2, 64), (2, 64, 128, 3, 1, 20), (1, 16, 40, 1, 1, 24), (3, 256, 200, 3, 2, 21),
]
big = [(32, 32, 64, 3, 2, 640), (32, 64, 128, 3, 2, 320), (32, 128, 256, 3, 2, 160), (32, 256, 512, 3, 2, 80), (32, 512, 1024, 3, 2, 40),
       (32, 256, 128, 1, 1, 80), (32, 512, 256, 1, 1, 40), (32, 1024, 512, 1, 1, 20), (32, 128, 64, 1, 1, 160), (32, 256, 255 + 1, 1, 1, 80)]
g = torch.Generator().manual_seed(0)


def case(B, Cin, Cout, k, s, H, check64=True, reps=0):
    x = torch.randn(B, Cin, H, H, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    scale, shift = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.1
    p = (k - 1) // 2
    xd = x.to(dev).contiguous(memory_format=torch.channels_last)
    wd = w.permute(0, 2, 3, 1).contiguous().to(dev)
    w3 = ops.split_bf16(wd)
    args = (xd, wd, scale.to(dev), shift.to(dev), k, s, (p, p, p, p), ops.ACT_LEAKY)
    y32 = ops.conv2d(*args)
    y3 = ops.conv2d(*args, b3=w3, b3_min_rows=1)
    torch.cuda.synchronize()
    d = (y3 - y32).abs().max().item()
    line = f'{B}x{Cin}->{Cout} k{k}s{s} {H}^2: |b3 - f32| {d:.2e} (max|y| {y32.abs().max().item():.2f})'
    if check64:
        ref = F.conv2d(F.pad(x, (p, p, p, p)).double(), w.double(), None, s) * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)
        ref = F.leaky_relu(ref, 0.1)
        e3 = (y3.cpu().double() - ref).abs().max().item() / ref.abs().max().item()
        e32 = (y32.cpu().double() - ref).abs().max().item() / ref.abs().max().item()
        line += f' | vs float64 / max|y|: b3 {e3:.2e}  f32 {e32:.2e}'
        assert e3 < 2e-5, line
    if reps:
        for fn, tag in ((lambda: ops.conv2d(*args), 'f32'), (lambda: ops.conv2d(*args, b3=w3, b3_min_rows=1), 'b3')):
            for _ in range(3):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            fl = 2.0 * B * (H // s) ** 2 * Cout * k * k * Cin
            line += f' | {tag} {ms:.4f} ms {fl / ms / 1e9:.0f} TF'
    print(line, flush=True)


import os
if int(os.environ.get('MYDET_B3_PF', '0')) < 10:          # (diagnostic instances >= 10 compute garbage on purpose)
    for sh in shapes:
        case(*sh)
for sh in big:
    case(*sh, check64=False, reps=20)
