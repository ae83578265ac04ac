"""Which Python lines of a forward issue aten::copy_ / clone / contiguous / cat (device-to-device copies)?
    python tools/find_copies.py [config] [batch] [size]"""
import collections, os, sys, traceback, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mydetection_amd import synth
from mydetection_amd.models.general import name_to_model
from mydetection_amd.utils.structures import batched_post_process
name = sys.argv[1] if len(sys.argv) > 1 else 'efficientdet-d1'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
S = int(sys.argv[3]) if len(sys.argv) > 3 else 640
model, cfg = name_to_model(name)
model.load_state_dict(synth.make_state_dict(model.state_dict(), name), strict=True)
model = model.eval().cuda()
x = synth.make_image_set(0, B, S, cfg['general.input_format']).cuda()
with torch.no_grad():
    for _ in range(2):
        bb, ci, sc = model.forward_candidates(x)
        batched_post_process(bb, ci, sc, 0.05, 0.5)
sites = collections.Counter()
allf = collections.Counter()
from torch.utils._python_dispatch import TorchDispatchMode
class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        n = str(func)
        allf[n] += 1
        if any(k in n for k in ('copy_', 'clone', 'cat', '_to_copy', 'contiguous', 'mul', 'add', 'zeros', 'fill', 'index', 'sigmoid', 'mean')):
            st = [f for f in traceback.extract_stack() if 'mydetection_amd' in f.filename]
            sites[(n, '%s:%d' % (os.path.relpath(st[-1].filename), st[-1].lineno) if st else '?')] += 1
        return func(*args, **(kwargs or {}))
with torch.no_grad(), Spy():
    bb, ci, sc = model.forward_candidates(x)
    batched_post_process(bb, ci, sc, 0.05, 0.5)
print('all aten calls:', allf.most_common(30))
for (n, where), c in sites.most_common(40):
    print(f'{c:4d}  {n:40s} {where}')
