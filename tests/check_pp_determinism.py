"""Diagnostic: repeat batched post-processing on one set of candidates and compare runs / the CPU oracle."""
import contextlib, io, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mydetection_amd import ops, synth
from mydetection_amd.models.general import name_to_model
from oracle import postprocess as pp
with contextlib.redirect_stdout(io.StringIO()):
    m, cfg = name_to_model('yolov3_80')
m.load_state_dict(synth.make_state_dict(m.state_dict(), 'yolov3_80')); m = m.eval().cuda()
x = synth.make_images(32, 640, seed=0).cuda()
with torch.no_grad():
    bb, ci, sc = m.forward_candidates(x)
runs = []
for r in range(6):
    rec = ops.postprocess(bb, ci, sc, 0.005, 0.45)
    torch.cuda.synchronize()
    runs.append((rec['count'].cpu().numpy().copy(), rec['index'].cpu().numpy().copy()))
    print('run', r, 'counts[:8]', runs[-1][0][:8], 'sum', runs[-1][0].sum())
for i in (0, 5):
    ob, oc, os_, src = pp.post_process(bb[i].cpu().numpy(), ci[i].cpu().numpy(), sc[i].cpu().numpy(), 0.005, 0.45)
    print('image', i, 'oracle kept', len(src), 'gpu kept', [int(c[i]) for c, _ in runs])
