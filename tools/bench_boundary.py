"""Cost of a dependent launch: 200 launches of a trivially small kernel (max pool of a 2x2 map) in one hipGraph, and the same
issued eagerly.   python tools/bench_boundary.py      (MI355X, round 3: 2.0 us per launch in a graph, 14 us eager)"""
import torch, sys, os
sys.path.insert(0, os.getcwd())
from mydetection_amd import ops
dev = torch.device('cuda')
x = torch.randn(1, 2, 2, 4, device=dev).permute(0, 3, 1, 2)
def run():
    return ops.maxpool3s2(x)
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    for _ in range(3): run()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=side):
    for _ in range(200): run()
g.replay(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): g.replay()
e1.record(); torch.cuda.synchronize()
print('dependent tiny kernels in a graph: %.2f us per launch' % (e0.elapsed_time(e1) / 1000 * 1e3))
# same, eager on one stream
e0.record()
for _ in range(1000): run()
e1.record(); torch.cuda.synchronize()
print('eager: %.2f us per launch' % (e0.elapsed_time(e1) / 1000 * 1e3))
