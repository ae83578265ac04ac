"""One split-bf16 conv layer, a few launches (driver for rocprofv3 --pmc passes): python tools/r05_b3_one.py B Cin Cout k s H"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mydetection_amd import ops
B, Cin, Cout, k, s, H = (int(v) for v in sys.argv[1:7])
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
x = torch.randn(B, Cin, H, H, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
w = (torch.randn(Cout, k, k, Cin, generator=g) / (Cin * k * k) ** 0.5).to(dev)
sc, sh = (torch.rand(Cout, generator=g) + 0.5).to(dev), (torch.randn(Cout, generator=g) * 0.1).to(dev)
w3 = ops.split_bf16(w)
p = (k - 1) // 2
for _ in range(4):
    ops.conv2d(x, w, sc, sh, k, s, (p, p, p, p), ops.ACT_LEAKY, b3=w3, b3_min_rows=1)
torch.cuda.synchronize()
