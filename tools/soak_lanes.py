"""Soak: SOAK (default 40) replays of the 1- and 2-lane hipGraphs of a configuration give the same bits every time and equal the
host-launched decomposition.  (Found the round-5 squeeze-excite tail problem: python tools/soak_lanes.py d1_fcs2_atss 32)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mydetection_amd import synth, ops
from mydetection_amd.graph import GraphedPath
from mydetection_amd.models.general import name_to_model
name = sys.argv[1] if len(sys.argv) > 1 else 'efficientdet-d1'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
m, cfg = name_to_model(name)
m.load_state_dict(synth.make_state_dict(m.state_dict(), name), strict=True)
m = m.eval().cuda()
x = synth.make_normalized_images(B, 640, seed=13).cuda()
for lanes in (1, 2):
    g = GraphedPath(m, x, 0.005, 0.5, lanes=lanes)
    g()
    torch.cuda.synchronize()
    ref = [t.clone() for t in g.cand]
    worst, nbad = 0.0, 0
    N = int(os.environ.get('SOAK', '40'))
    for r in range(N):
        g()
        torch.cuda.synchronize()
        d = float((g.cand[2] - ref[2]).abs().max())
        worst = max(worst, d); nbad += d > 0
    rec = {k: v.clone() for k, v in g().items()}
    torch.cuda.synchronize()
    eag = g.eager()
    torch.cuda.synchronize()
    same = all(torch.equal(rec[k], eag[k]) for k in rec)
    print(name, B, 'lanes', lanes, 'replays differing from the first', nbad, 'of', N, 'worst score diff', worst, '| replay == host-launched decomposition:', same, flush=True)
