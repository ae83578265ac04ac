"""Time mydet_decode_levels_f32 alone on the three head layouts at bench sizes (HBM roofline check).

    python tools/bench_decode.py [--batch 32] [--size 640] [--iters 200]

Prints one line per layout: average launch time (HIP events on the launch stream), algorithmic bytes, GB/s and the
fraction of the 8 TB/s HBM peak.
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mydetection_amd import ops  # noqa: E402


def layouts(B, S):
    dev = 'cuda'
    g = torch.Generator(device=dev).manual_seed(0)
    rnd = lambda *s: torch.randn(*s, device=dev, generator=g)
    out = {}
    # YOLOv3: one [B,H,W,255] tensor per level, anchor-major channels (85 per anchor)
    lv, n = [], 0
    for st in (8, 16, 32):
        H = S // st
        t = rnd(B, H, H, 256)[..., :255]          # ld 256 keeps rows 16-byte aligned like the product's head conv
        lv.append(dict(box=t, ldbox=256, cls=t, ldcls=256, anchors_wh=[[10, 13], [16, 30], [33, 23]], H=H, W=H,
                       stride=st, n_off=n))
        n += 3 * H * H
    out['yolo'] = (ops.DECODE_YOLO, lv, dict(box_astride=85, box_c0=0, cls_astride=85, cls_c0=5, conf_c0=4, A=3, C=80), n)
    # RetinaNet head: box [B,H,W,36] and cls [B,H,W,720] per level, 9 anchors
    lv, n = [], 0
    for st in (8, 16, 32, 64, 128):
        H = (S + st - 1) // st
        bx, cl = rnd(B, H, H, 36), rnd(B, H, H, 720)
        lv.append(dict(box=bx, ldbox=36, cls=cl, ldcls=720, anchors_wh=[[32 * st / 8, 32 * st / 8]] * 9, H=H, W=H,
                       stride=st, n_off=n))
        n += 9 * H * H
    out['retina'] = (ops.DECODE_RETINA, lv, dict(box_astride=4, box_c0=0, cls_astride=80, cls_c0=0, conf_c0=0, A=9, C=80), n)
    # FCOS/ATSS head: box [B,H,W,4], cls [B,H,W,84] (80 classes + centerness at 80), 1 anchor
    lv, n = [], 0
    for st in (8, 16, 32, 64, 128):
        H = (S + st - 1) // st
        bx, cl = rnd(B, H, H, 4), rnd(B, H, H, 84)
        lv.append(dict(box=bx, ldbox=4, cls=cl, ldcls=84, anchors_wh=None, H=H, W=H, stride=st, n_off=n))
        n += H * H
    out['fcos'] = (ops.DECODE_FCOS, lv, dict(box_astride=4, box_c0=0, cls_astride=84, cls_c0=0, conf_c0=80, A=1, C=80), n)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--size', type=int, default=640)
    ap.add_argument('--iters', type=int, default=200)
    ap.add_argument('--only', default='')
    a = ap.parse_args()
    B, S = a.batch, a.size
    for name, (mode, lv, kw, N) in layouts(B, S).items():
        if a.only and name != a.only:
            continue
        bbox = torch.empty(B, N, 4, device='cuda')
        ci = torch.empty(B, N, dtype=torch.int64, device='cuda')
        sc = torch.empty(B, N, device='cuda')
        run = lambda: ops.decode_levels(mode, lv, kw['box_astride'], kw['box_c0'], kw['cls_astride'], kw['cls_c0'],
                                        kw['conf_c0'], kw['A'], kw['C'], B, (S, S), bbox, ci, sc)
        # the host side of a launch (ctypes marshalling of the level table) costs more than the kernel: replay a
        # hipGraph of `reps` back-to-back launches so the events see device time only
        reps = 20
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            for _ in range(3):
                run()
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            for _ in range(reps):
                run()
        graph.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = max(1, a.iters // reps)
        e0.record()
        for _ in range(n):
            graph.replay()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / (n * reps)
        per_pix = kw['A'] * (kw['C'] + 4 + (0 if mode == ops.DECODE_RETINA else 1))
        nbytes = sum(4.0 * B * l['H'] * l['W'] * per_pix + 28.0 * B * kw['A'] * l['H'] * l['W'] for l in lv)
        gbs = nbytes / ms / 1e6
        print(f'{name:7s} B={B} S={S} N={N} {ms * 1e3:8.1f} us  {nbytes / 1e6:8.1f} MB  {gbs:7.0f} GB/s  '
              f'{gbs / 8000:.3f} of HBM peak', flush=True)


if __name__ == '__main__':
    main()
