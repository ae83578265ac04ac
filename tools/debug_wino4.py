"""Debug helper for the F(4x4,3x3) kernel: max error per (image, channel block) of one shape against float64.
    python tools/debug_wino4.py B Cin Cout H W [res]"""
import os, sys, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mydetection_amd import ops
dev = torch.device('cuda')
B, Cin, Cout, H, W = map(int, sys.argv[1:6])
res = 'res' in sys.argv[6:]
act = 1 if 'leaky' in sys.argv[6:] else 0
g = torch.Generator().manual_seed(1)
x = torch.randn(B, Cin, H, W, generator=g); w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** .5
r = torch.randn(B, Cout, H, W, generator=g) if res else None
ref = F.conv2d(x.double(), w.double(), padding=1)
if act:
    ref = torch.where(ref > 0, ref, ref * 0.1)
ref = ref + (r.double() if res else 0)
wd = w.permute(0, 2, 3, 1).contiguous().to(dev)
u4 = ops.wino4_weights(wd)
for rep in range(3):
    y = ops.conv2d(x.to(dev).contiguous(memory_format=torch.channels_last), wd, None, torch.zeros(Cout, device=dev), 3, 1, (1, 1, 1, 1), act,
                   residual=r.to(dev).contiguous(memory_format=torch.channels_last) if res else None, wino4=u4)
    e = (y.cpu().double() - ref).abs()
    print('rep', rep, 'max err', e.max().item(), 'ref max', ref.abs().max().item())
    if e.max() > 1e-3:
        bad = (e > 1e-3)
        print(' bad fraction', bad.float().mean().item())
        print(' per image:', e.amax(dim=(1, 2, 3)).tolist())
        cb = e.amax(dim=(0, 2, 3)).view(-1, 16).amax(dim=1)
        print(' per 16-channel block:', [round(v, 4) for v in cb.tolist()])
        print(' per column:', [round(v, 3) for v in e.amax(dim=(0, 1, 2)).tolist()])
        idx = bad.nonzero()
        for b_, c_, h_, w_ in idx[:24].tolist():
            print('   bad at img %d ch %d y %d x %d: got %.5f ref %.5f' % (b_, c_, h_, w_, y[b_, c_, h_, w_].item(), ref[b_, c_, h_, w_].item()))
        print(' per row:', [round(v, 3) for v in e.amax(dim=(0, 1, 3)).tolist()])
