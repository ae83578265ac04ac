"""Round 6: conv_p3_kernel (LDS-resident patch, split-bf16) against the shipped path of the same layer, back to back in one process:
the five stride-2 3x3 layers of Darknet-53 and the 32 -> 64 stride-1 layer of the first DarkBlock at batch 32, 640 x 640.
    python tools/r06/bench_p3.py [reps]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mydetection_amd import ops

dev = torch.device('cuda:0')
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
B = int(os.environ.get('P3_BATCH', '32'))
LAYERS = [(32, 64, 2, 640), (64, 128, 2, 320), (128, 256, 2, 160), (256, 512, 2, 80), (512, 1024, 2, 40), (32, 64, 1, 320)]
if os.environ.get('P3_STRIPS'):
    LAYERS = [(256, 512, 2, 80), (512, 1024, 2, 40), (128, 256, 2, 160)]
if os.environ.get('P3_S1'):
    LAYERS = [(64, 128, 1, 160), (128, 256, 1, 80), (32, 64, 1, 320)]


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


g = torch.Generator().manual_seed(1)
for Cin, Cout, s, H in LAYERS:
    x = torch.randn(B, Cin, H, H, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Cout, 3, 3, Cin, generator=g) / (9 * Cin) ** 0.5).to(dev)
    sc, sh = (torch.rand(Cout, generator=g) + 0.5).to(dev), (torch.randn(Cout, generator=g) * 0.1).to(dev)
    res = None
    Ho = (H - 1) // s + 1
    if s == 1:
        res = torch.randn(B, Cout, Ho, Ho, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    w3 = ops.split_bf16(w)
    u = ops.wino_weights(w) if s == 1 else None
    u4 = ops.wino4_weights(w) if s == 1 and Cin >= 64 else None
    y_p3 = ops.conv3x3_p3(x, w3, sc, sh, s, ops.ACT_LEAKY, residual=res)
    if s == 1:
        base = lambda: ops.conv2d(x, w, sc, sh, 3, 1, (1, 1, 1, 1), ops.ACT_LEAKY, residual=res, wino=u, wino4=u4)
        what = 'conv_wino4 F(4x4)' if u4 is not None else 'conv_wino F(2x2)'
    else:
        base = lambda: ops.conv2d(x, w, sc, sh, 3, s, (1, 1, 1, 1), ops.ACT_LEAKY, b3=w3)
        what = 'conv_igemm_b3'
    ops.CONV_P3 = False                      # the baseline is the shipped path WITHOUT the patch-resident kernel
    y_b = base()
    d = (y_p3 - y_b).abs().max().item() / y_b.abs().max().item()
    p3 = lambda: ops.conv3x3_p3(x, w3, sc, sh, s, ops.ACT_LEAKY, residual=res)
    tb, tp = [], []
    for _ in range(3):                       # alternate: the chip's clock drifts with what ran before
        tb.append(timed(base))
        tp.append(timed(p3))
    ops.CONV_P3 = True
    t_b, t_p = sorted(tb)[1], sorted(tp)[1]
    fl = 2.0 * B * Ho * Ho * Cout * 9 * Cin
    print(f'{Cin:4d}->{Cout:4d} k3s{s} @{H:3d}^2 batch {B}: {what:16s} {t_b:.4f} ms ({fl / t_b / 1e9:6.1f} TFLOP/s)   conv_p3 {t_p:.4f} ms ({fl / t_p / 1e9:6.1f} TFLOP/s)   '
          f'x{t_b / t_p:.2f}   max rel diff {d:.1e}   runs {["%.3f" % v for v in tb]} / {["%.3f" % v for v in tp]}', flush=True)
    del x, w, y_p3, y_b, res
