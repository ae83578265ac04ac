"""Soak: N eager full-batch forwards of a configuration on ONE stream; every launch's output compared with the first pass (clones),
the first launch whose output differs is reported.  python tools/soak_eager.py efficientdet-d1 16 [N]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mydetection_amd import synth, ops
from mydetection_amd.models.general import name_to_model
name = sys.argv[1] if len(sys.argv) > 1 else 'efficientdet-d1'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
N = int(sys.argv[3]) if len(sys.argv) > 3 else 200
m, cfg = name_to_model(name)
m.load_state_dict(synth.make_state_dict(m.state_dict(), name), strict=True)
m = m.eval().cuda()
x = synth.make_normalized_images(B, 640, seed=13).cuda()
log = []
def wrap(fname):
    f = getattr(ops, fname)
    def g(*a, **k):
        r = f(*a, **k)
        outs = r if isinstance(r, tuple) else (r,)
        for j, t in enumerate(outs):
            if torch.is_tensor(t) and not (fname in ('dwconv', 'mbconv_expand_dw', 'stem_dw') and j == 1 and t.dim() == 3):   # (squeeze partials: slot S is scratch)
                log.append((f'{fname}[{j}] {tuple(t.shape)} b3={k.get("b3") is not None} se={k.get("se") is not None} gate={k.get("gate") is not None}', t.clone()))
        return r
    setattr(ops, fname, g)
for n in ('conv2d', 'dwconv', 'se_gate', 'mbconv_expand_dw', 'stem_dw', 'sepconv_nodes', 'pw_skinny', 'conv2d_stem'):
    if hasattr(ops, n):
        wrap(n)
with torch.no_grad():
    ref_out = m.forward_candidates(x)
torch.cuda.synchronize()
ref = list(log)
bad_runs = 0
first_bad = {}
for r in range(N):
    log.clear()
    with torch.no_grad():
        out = m.forward_candidates(x)
    torch.cuda.synchronize()
    same = all(torch.equal(a, b) for a, b in zip(out, ref_out))
    if not same:
        bad_runs += 1
        for i, ((nm, t), (_, t0)) in enumerate(zip(log, ref)):
            if not torch.equal(t, t0):
                d = (t - t0).abs()
                first_bad[i] = first_bad.get(i, 0) + 1
                if first_bad[i] == 1:
                    idx = (d > 0).nonzero()
                    print(f'run {r}: first differing launch {i}/{len(ref)} {nm}: {int((d > 0).sum())} elements, max {float(d.max()):.3e}, first at {idx[0].tolist()} last {idx[-1].tolist()}', flush=True)
                    if i > 0:
                        print('     previous launch:', ref[i - 1][0], flush=True)
                break
print(name, B, 'runs differing from the first:', bad_runs, 'of', N, 'by first differing launch:', first_bad)
