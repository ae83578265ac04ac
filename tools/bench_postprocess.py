"""Time mydet_postprocess_f32 alone on the candidates of one YOLOv3-80 forward (batch 32, 640x640): hipGraph replay
of 20 launches, HIP events.  Prints the per-image passing / kept counts and the time per launch.
    python tools/bench_postprocess.py [--batch 32] [--size 640] [--conf 0.005] [--nms 0.45]"""
import argparse
import contextlib
import io
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mydetection_amd import ops, synth                                    # noqa: E402
from mydetection_amd.models.general import name_to_model                 # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=32)
ap.add_argument('--size', type=int, default=640)
ap.add_argument('--conf', type=float, default=0.005)
ap.add_argument('--nms', type=float, default=0.45)
ap.add_argument('--config', default='yolov3_80')
a = ap.parse_args()
with contextlib.redirect_stdout(io.StringIO()):
    model, cfg = name_to_model(a.config)
model.load_state_dict(synth.make_state_dict(model.state_dict(), a.config))
model = model.eval().cuda()
x = (synth.make_images if cfg['general.input_format'] == 'RGB_1' else synth.make_normalized_images)(a.batch, a.size, seed=0).cuda()
with torch.no_grad():
    bb, ci, sc = model.forward_candidates(x)
npass = (sc >= a.conf).sum(dim=1)
run = lambda: ops.postprocess(bb, ci, sc, a.conf, a.nms)
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    for _ in range(3):
        rec = run()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=side):
    for _ in range(20):
        run()
g.replay()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    g.replay()
e1.record()
torch.cuda.synchronize()
torch.cuda.synchronize()
rec = run()
again = run()
torch.cuda.synchronize()
assert os.environ.get('MYDET_PP_STOP') or (torch.equal(rec['count'], again['count']) and torch.equal(rec['index'], again['index']))
print(f'{a.config} B={a.batch} N={sc.shape[1]} pass conf: mean {npass.float().mean():.0f} max {int(npass.max())}; '
      f'kept mean {rec["count"].float().mean():.0f}; {e0.elapsed_time(e1) / 200 * 1e3:.1f} us per launch')
