"""Where the post-process launch spends its time on the EfficientDet-D1 shape (8 images x 76 725 candidates, one lane):
conf above every score = the score sweep alone; few pass = sweep + sort + NMS; everything passes = + the top-k search.
Answers whether keys emitted by the fused decode (no sweep in the post-process) would pay.  Usage: python tools/pp_phases.py"""
import json
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mydetection_amd import ops


def timed(fn, n=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    dev = torch.device('cuda:0')
    out = {}
    for name, B, N in (('d1_lane', 8, 76725), ('yolov3_b32', 32, 25200), ('yolov3_b1_512', 1, 16128)):
        g = torch.Generator().manual_seed(3)
        cxcy = torch.rand(B, N, 2, generator=g) * 640
        wh = torch.rand(B, N, 2, generator=g) * 80 + 8
        bbox = torch.cat((cxcy, wh), 2).to(dev)
        cls = torch.randint(0, 80, (B, N), generator=g).to(dev)
        score = torch.rand(B, N, generator=g).to(dev)
        rec = torch.empty((B, ops._lib.REC_WORDS), dtype=torch.int32, device=dev)
        row = {}
        for label, conf in (('sweep_only(nothing passes)', 2.0), ('~100 pass', 1.0 - 100.0 / N), ('~512 pass', 1.0 - 512.0 / N),
                            ('~4000 pass', 1.0 - 4000.0 / N), ('all pass', 0.0)):
            row[label] = round(timed(lambda: ops.postprocess(bbox, cls, score, conf, 0.45, records=rec)), 2)
        out[name] = row
    print(json.dumps({'postprocess_us_per_launch': out}))


if __name__ == '__main__':
    main()
