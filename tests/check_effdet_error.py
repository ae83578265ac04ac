"""Diagnostic: error of the HIP EfficientDet path vs the CPU oracle in fp32 and fp64 (per stage)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mydetection_amd import synth
from mydetection_amd.models.general import name_to_model
from oracle import efficientdet as oe

name = sys.argv[1] if len(sys.argv) > 1 else 'efficientdet-d1'
size = int(sys.argv[2]) if len(sys.argv) > 2 else 640
m, cfg = name_to_model(name)
m.load_state_dict(synth.make_state_dict(m.state_dict(), name))
m = m.eval().cuda()
x = synth.make_normalized_images(2, size, seed=7)
sd = {k: v.cpu() for k, v in m.state_dict().items()}
sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
atss = name == 'd1_fcs2_atss'
with torch.no_grad():
    c32 = oe.backbone(x, sd, c6c7='conv' if atss else 'maxpool'); p32 = oe.bifpn(c32, sd); h32 = oe.head(p32, sd)
    c64 = oe.backbone(x.double(), sd64, c6c7='conv' if atss else 'maxpool'); p64 = oe.bifpn(c64, sd64); h64 = oe.head(p64, sd64)
    cg = m.backbone(x.cuda()); pg = m.fpn(cg); rg = m.rpn(pg)
def rel(a, b):
    return ((a.double() - b.double()).abs().max() / b.double().abs().max()).item()
for lvl in range(5):
    print(f'L{lvl} backbone: gpu-vs-64 {rel(cg[lvl].cpu(), c64[lvl]):.2e}  cpu32-vs-64 {rel(c32[lvl], c64[lvl]):.2e} | '
          f'fpn: gpu {rel(pg[lvl].cpu(), p64[lvl]):.2e} cpu32 {rel(p32[lvl], p64[lvl]):.2e} | '
          f'cls head: gpu {rel(rg[lvl].packed["cls"][0].cpu(), h64[lvl][0]):.2e} cpu32 {rel(h32[lvl][0], h64[lvl][0]):.2e} '
          f'absmax logit {h64[lvl][0].abs().max().item():.1f} abs err gpu {(rg[lvl].packed["cls"][0].cpu().double()-h64[lvl][0]).abs().max().item():.2e} cpu32 {(h32[lvl][0].double()-h64[lvl][0]).abs().max().item():.2e}')
