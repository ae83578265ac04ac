"""Board power and shader clock (rocm-smi, sampled twice a second) while ONE layer of the headline runs back to back for ~4 s:
energy per launch = power x time per launch.  The headline step runs at the board's power cap (profiles/r05_power_headline.txt), so
a layer's share of the step's ENERGY is what decides the step time."""
import os, re, subprocess, sys, threading, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mydetection_amd import ops
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)


def mk(B, C, H):
    return torch.randn(B, C, H, H, generator=g).to(dev).contiguous(memory_format=torch.channels_last)


def layer(kind, B, Cin, Cout, k, s, H):
    x = mk(B, Cin, H)
    w = (torch.randn(Cout, k, k, Cin, generator=g) / (Cin * k * k) ** 0.5).to(dev)
    sc, sh = (torch.rand(Cout, generator=g) + 0.5).to(dev), (torch.randn(Cout, generator=g) * 0.1).to(dev)
    p = (k - 1) // 2
    res = mk(B, Cout, H // s) if kind in ('wino4', 'wino') else None
    kw = {}
    if kind == 'b3':
        kw = dict(b3=ops.split_bf16(w), b3_min_rows=1)
    elif kind == 'wino4':
        kw = dict(wino4=ops.wino4_weights(w), residual=res)
    elif kind == 'wino':
        kw = dict(wino=ops.wino_weights(w), residual=res)
    return lambda: ops.conv2d(x, w, sc, sh, k, s, (p, p, p, p), ops.ACT_LEAKY, **kw)


def smi():
    out = subprocess.run(['rocm-smi', '--showpower', '--showclocks'], capture_output=True, text=True).stdout
    pw = re.search(r'Power \(W\): ([\d.]+)', out)
    ck = re.search(r'sclk clock level: \d+: \((\d+)Mhz\)', out)
    return (float(pw.group(1)) if pw else float('nan'), int(ck.group(1)) if ck else 0)


cases = [('F(4x4) 128->256 @80^2 (+skip)', layer('wino4', 32, 128, 256, 3, 1, 80), 120.8),
         ('F(4x4) 512->1024 @20^2 (+skip)', layer('wino4', 32, 512, 1024, 3, 1, 20), 120.8),
         ('split-bf16 128->256 s2 @160^2', layer('b3', 32, 128, 256, 3, 2, 160), 120.8),
         ('float32   128->256 s2 @160^2', layer('f32', 32, 128, 256, 3, 2, 160), 120.8),
         ('split-bf16 256->128 1x1 @80^2', layer('b3', 32, 256, 128, 1, 1, 80), 13.42),
         ('float32   256->128 1x1 @80^2', layer('f32', 32, 256, 128, 1, 1, 80), 13.42),
         ('F(2x2) 32->64 @320^2 (+skip)', layer('wino', 32, 32, 64, 3, 1, 320), 120.8),
         ('float32 64->32 1x1 @320^2', layer('f32', 32, 64, 32, 1, 1, 320), 13.42)]
print(f'{"layer":34s} {"ms/launch":>9s} {"W":>7s} {"sclk MHz":>9s} {"J/launch":>9s} {"GFLOP/J":>8s}')
for name, fn, gflop in cases:
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    samples, stop = [], False

    def sampler():
        while not stop:
            samples.append(smi())
            time.sleep(0.4)
    th = threading.Thread(target=sampler)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 0
    t_end = time.time() + 4.0
    e0.record()
    th.start()
    while time.time() < t_end:
        for _ in range(50):
            fn()
        torch.cuda.synchronize()
        n += 50
    e1.record(); e1.synchronize()
    stop = True
    th.join()
    ms = e0.elapsed_time(e1) / n
    s = samples[len(samples) // 3:]                       # after the first third: settled
    pw = sum(a for a, _ in s) / len(s)
    ck = sum(b for _, b in s) / len(s)
    print(f'{name:34s} {ms:9.4f} {pw:7.0f} {ck:9.0f} {pw * ms / 1e3:9.4f} {gflop / (pw * ms / 1e3):8.1f}', flush=True)
    time.sleep(1.0)
