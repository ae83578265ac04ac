"""One layer on conv_p3_kernel, a few launches (for rocprofv3 --pmc passes: tools/r06/pmc_p3.sh).  args: B Cin Cout stride H"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mydetection_amd import ops
B, Cin, Cout, s, H = (int(v) for v in sys.argv[1:6])
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(1)
x = torch.randn(B, Cin, H, H, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
w = (torch.randn(Cout, 3, 3, Cin, generator=g) / (9 * Cin) ** 0.5).to(dev)
sc, sh = (torch.rand(Cout, generator=g) + 0.5).to(dev), (torch.randn(Cout, generator=g) * 0.1).to(dev)
w3 = ops.split_bf16(w)
for _ in range(5):
    y = ops.conv3x3_p3(x, w3, sc, sh, s, ops.ACT_LEAKY)
torch.cuda.synchronize()
print('ok', tuple(y.shape))
