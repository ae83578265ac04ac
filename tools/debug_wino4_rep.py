"""Debug helper: back-to-back launches of the F(4x4,3x3) pair on one shape; where do runs differ?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mydetection_amd import ops
dev = torch.device('cuda')
B, Cin, Cout, H, W = map(int, sys.argv[1:6])
g = torch.Generator().manual_seed(5)
x = torch.randn(B, Cin, H, W, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
w = (torch.randn(Cout, 3, 3, Cin, generator=g) / (Cin * 9) ** 0.5).to(dev)
shift = (torch.randn(Cout, generator=g) * 0.1).to(dev)
res = torch.randn(B, Cout, H, W, generator=g).to(dev).contiguous(memory_format=torch.channels_last) if 'res' in sys.argv else None
u4 = ops.wino4_weights(w)
direct = ops.conv2d(x, w, None, shift, 3, 1, (1, 1, 1, 1), 1, residual=res)
n = int(os.environ.get('REPS', '8'))
outs = [ops.conv2d(x, w, None, shift, 3, 1, (1, 1, 1, 1), 1, residual=res, wino4=u4) for _ in range(n)]
torch.cuda.synchronize()
for i, o in enumerate(outs):
    e = (o - direct).abs()
    bad = e > 1e-3
    print('run', i, 'max diff vs direct', e.max().item(), 'bad', int(bad.sum()))
    if bad.any():
        idx = bad.nonzero()
        print('   images', sorted(set(idx[:, 0].tolist()))[:10], 'channels', sorted(set(idx[:, 1].tolist()))[:20],
              'rows', sorted(set(idx[:, 2].tolist()))[:12], 'cols', sorted(set(idx[:, 3].tolist()))[:12])
