import sys, torch, torch.nn.functional as F
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mydetection_amd import ops
dev=torch.device('cuda')
def run(B,Cin,Cout,H,W,mode):
    g=torch.Generator().manual_seed(1)
    x=torch.randn(B,Cin,H,W,generator=g); w=torch.randn(Cout,Cin,3,3,generator=g)/(Cin*9)**.5
    if mode=='ones': x=torch.ones_like(x); w=torch.ones_like(w)/(Cin*9)
    if mode=='center':  # only centre tap
        w=torch.zeros_like(w); w[:,:,1,1]=torch.randn(Cout,Cin,generator=g)
    ref=F.conv2d(x.double(),w.double(),padding=1)
    wd=w.permute(0,2,3,1).contiguous().to(dev)
    u4=ops.wino4_weights(wd)
    y=ops.conv2d(x.to(dev).contiguous(memory_format=torch.channels_last),wd,None,torch.zeros(Cout,device=dev),3,1,(1,1,1,1),0,wino4=u4)
    e=(y.cpu().double()-ref).abs()
    print(mode,(B,Cin,Cout,H,W),'max err',e.max().item(),'ref max',ref.abs().max().item())
    if e.max()>1e-3 and H<=8 and Cout<=64:
        print(' y[0,0]:\n',y[0,0].cpu()); print(' ref[0,0]:\n',ref[0,0])
        print(' err per channel (first 8):', e.amax(dim=(0,2,3))[:8])
for mode in ('ones','center','rand'):
    run(1,4,64,4,4,mode)
run(1,4,64,8,8,'rand'); run(1,8,64,8,8,'rand'); run(2,4,128,8,8,'rand')
