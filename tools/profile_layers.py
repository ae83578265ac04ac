"""Per-layer timing of the YOLOv3-80 conv launches (HIP events), batch 32 640x640 by default.
    python tools/profile_layers.py [--batch 32] [--size 640] [--reps 5]
Prints one line per distinct conv shape: launches, ms per launch, TFLOP/s, fraction of FP32 MFMA peak."""
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
ap.add_argument('--reps', type=int, default=5)
ap.add_argument('--config', default='yolov3_80')
a = ap.parse_args()
with contextlib.redirect_stdout(io.StringIO()):
    model, cfg = name_to_model(a.config)
model.load_state_dict(synth.make_state_dict(model.state_dict(), a.config))
model = model.eval().cuda()
x = (synth.make_images if cfg['general.input_format'] == 'RGB_1' else synth.make_normalized_images)(a.batch, a.size, seed=0).cuda()
with torch.no_grad():
    for _ in range(2):
        model.forward_candidates(x)
    torch.cuda.synchronize()
    ops.TIMER, ops.TIMER_DETAIL = ops.KernelTimer(), True
    for _ in range(a.reps):
        model.forward_candidates(x)
    torch.cuda.synchronize()
summ = ops.TIMER.summary()
tot_ms = sum(v[1] for v in summ.values()) / a.reps
import re
print(f'{"kernel":44s} {"n/step":>6s} {"ms/launch":>10s} {"ms/step":>8s} {"%step":>6s} {"TFLOP/s":>8s} {"%peak":>6s} {"GB/s":>8s}')
for k, (n, ms, work) in sorted(summ.items(), key=lambda kv: -kv[1][1]):
    per = ms / n
    if k.startswith('conv_igemm') or k.startswith('conv_wino') or k.startswith('conv_p3'):
        tf = work / n / (per * 1e-3) / 1e12
        m = re.match(r'conv_(?:igemm|wino) (\d+)->(\d+) k(\d)s(\d) (\d+)x(\d+)', k)
        gbs = float('nan')
        if m:       # algorithmic bytes: input once + output once (+ weights)
            cin, cout, kk, st, h, w = map(int, m.groups())
            gbs = 4.0 * (a.batch * h * w * cin + a.batch * (h // st) * (w // st) * cout + cout * cin * kk * kk) / (per * 1e-3) / 1e9
    else:
        tf, gbs = float('nan'), work / n / (per * 1e-3) / 1e9
    print(f'{k:44s} {n / a.reps:6.0f} {per:10.4f} {ms / a.reps:8.3f} {100 * ms / a.reps / tot_ms:6.1f} {tf:8.1f} {100 * tf / 157.3:6.1f} {gbs:8.0f}')
print(f'total kernel ms/step {tot_ms:.3f}')
