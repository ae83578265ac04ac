"""Benchmark of the detection inference hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

One step = one pass of the hot path over one batch resident in HBM:
    images [32,3,640,640] -> Darknet-53 -> YOLOv3 FPN -> head -> decode -> conf filter (0.005)
    -> top-512 -> class-aware NMS (0.45) [-> all-gather of detection records when N>1]
Workload = BASELINE.json configs[1] (yolov3_80, batch 32 per GPU, 640x640, synthetic weights/images).
Metric: images/sec over all GPUs (weak scaling: 32 images per GPU).

The JSON line also carries
  roofline     -- the dominant kernel (the conv family with the most time per step: fused Winograd or
                  implicit GEMM, both on FP32 MFMA): algorithmic FLOPs of its launches / their HIP-event
                  durations measured inside the timed region, against the 157.3 TFLOP/s FP32 matrix
                  peak (MI355X_MICROARCH.md); `mfma_frac` = multiplies actually issued / peak
  cpu_baseline -- the CPU oracle (port of the reference path) timed on this host's cores on a
                  bounded sample (rank 0, N=1 only)
  stages       -- per-kernel-family time per step, incl. the NMS launch (latency-bound; p50 reported)
"""
import argparse
import json
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0


def cpu_baseline(cfg, size, sample_batch, repeats, max_threads=16):
    """Oracle forward + post-process on the host cores (reported baseline, not the target)."""
    from mydetection_amd import synth
    from mydetection_amd.models.general import state_dict_template
    from oracle import postprocess as opp, yolov3 as oy
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        pass
    cores = min(cores, max_threads)         # a 1-GPU box owns a 16-CPU share of the host, not all of it
    torch.set_num_threads(cores)
    sd = synth.make_state_dict(state_dict_template('yolov3_80'))
    x = synth.make_images(sample_batch, size, seed=0)

    def once():
        with torch.no_grad():
            bb, ci, sc = oy.forward(x, sd)
        for b in range(sample_batch):
            opp.post_process(bb[b].numpy(), ci[b].numpy(), sc[b].numpy(), cfg['test.ap_conf_thres'], cfg['test.nms_thres'])
    once()
    times = []
    for _ in range(repeats):
        t = time.perf_counter()
        once()
        times.append(time.perf_counter() - t)
    med = statistics.median(times)
    return {'value': round(sample_batch / med, 3), 'unit': 'images/sec', 'cores': cores, 'kind': 'port',
            'sample': f'oracle/ (torch-CPU restatement of the reference path + C NMS), yolov3_80 batch {sample_batch} '
                      f'{size}x{size}, forward + per-image post_process, median of {repeats} after 1 warm-up'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=32, help='images per GPU')
    ap.add_argument('--size', type=int, default=640)
    ap.add_argument('--config', default='yolov3_80', help='yolov3_80 (headline) | efficientdet-d1 | d1_fcs2_atss')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--graph', action='store_true', help='replay the step from a captured hipGraph')
    ap.add_argument('--cpu-sample-batch', type=int, default=2)
    ap.add_argument('--cpu-repeats', type=int, default=3)
    ap.add_argument('--cpu-threads', type=int, default=16)
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}'
    assert torch.cuda.is_available(), 'bench.py needs MI355X GPUs'
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    rehearse = world == 1 and 'RANK' in os.environ and os.environ.get('MYDET_REHEARSE_RCCL') == '1'
    if world > 1 or rehearse:      # (rehearse: one-rank RCCL group, exercises the collective path on a 1-GPU box)
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', device_id=dev)          # nccl == RCCL on ROCm

    from mydetection_amd import _lib, ops, parallel, synth
    from mydetection_amd.models.general import name_to_model
    from mydetection_amd.utils.structures import batched_post_process
    _lib.lib()                                                   # loud failure if the HIP library is missing

    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        model, cfg = name_to_model(args.config)
    model.load_state_dict(synth.make_state_dict(model.state_dict(), args.config), strict=True)
    model = model.eval().to(dev)
    conf, nms = cfg['test.ap_conf_thres'], cfg['test.nms_thres']
    # each rank owns its shard of the global batch; resident in HBM before the timed region
    make = synth.make_images if cfg['general.input_format'] == 'RGB_1' else synth.make_normalized_images
    x = make(args.batch, args.size, seed=rank).to(dev)

    graphed = None
    if args.graph:
        from mydetection_amd.graph import GraphedPath
        graphed = GraphedPath(model, x, conf, nms)

    def step():
        if graphed is not None:
            return parallel.gather_detections(graphed(), always=rehearse)
        with torch.no_grad():
            bb, ci, sc = model.forward_candidates(x)
            rec = batched_post_process(bb, ci, sc, conf, nms)
            return parallel.gather_detections(rec, always=rehearse)

    def barrier():
        if world > 1 or rehearse:
            torch.distributed.barrier()

    for _ in range(args.warmup):
        rec = step()
    torch.cuda.synchronize()

    if graphed is None:
        ops.TIMER = ops.KernelTimer()                            # HIP events on the launch stream
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        rec = step()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    timer, ops.TIMER = ops.TIMER, None
    if timer is None:        # graph replay hides the launches from the event timer: price the kernels eagerly, after
        ops.TIMER = ops.KernelTimer()
        with torch.no_grad():
            for _ in range(args.steps):
                bb, ci, sc = model.forward_candidates(x)
                batched_post_process(bb, ci, sc, conf, nms)
        torch.cuda.synchronize()
        timer, ops.TIMER = ops.TIMER, None

    if world > 1 or rehearse:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())

    summ = timer.summary()
    if args.config != 'yolov3_80':          # secondary configs: per-kernel-family table only
        if rank == 0:
            tot = sum(v[1] for v in summ.values()) / args.steps
            print(json.dumps({'metric': f'images/sec, {args.config}, batch {args.batch}/GPU, {args.size}x{args.size}',
                              'value': round(world * args.batch * args.steps / elapsed, 2), 'unit': 'images/sec',
                              'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
                              'ms_per_step': round(elapsed / args.steps * 1e3, 3), 'kernel_ms_per_step': round(tot, 3),
                              'dtype': 'f32', 'data': 'synthetic',
                              'stages': {k: {'launches_per_step': v[0] / args.steps, 'ms_per_step': round(v[1] / args.steps, 4),
                                             'work_per_s': round(v[2] / (v[1] * 1e-3) / 1e9, 1)} for k, v in summ.items()}}))
        if world > 1:
            torch.distributed.destroy_process_group()
        return
    stages = {k: {'launches_per_step': v[0] / args.steps, 'ms_per_step': round(v[1] / args.steps, 4)} for k, v in summ.items()}
    pp_spans = [a.elapsed_time(b) for a, b, _ in timer.spans['postprocess']]
    stages['postprocess']['p50_ms'] = round(statistics.median(pp_spans), 4)
    stages['postprocess']['mean_detections_per_image'] = round(float(rec['count'].float().mean()), 1)
    dec_ms, dec_bytes = summ['decode'][1], summ['decode'][2]
    stages['decode']['achieved_GBs'] = round(dec_bytes / (dec_ms * 1e-3) / 1e9, 1)
    stages['decode']['frac_hbm_peak'] = round(dec_bytes / (dec_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)
    # Dominant kernel = the conv family with the most time per step.  `achieved` prices its launches with the
    # ALGORITHMIC flops of the layers (2*MACs of the direct form); the Winograd kernel issues 2.25x
    # fewer multiplies than that, so its matrix-pipe occupancy is reported next to it as `mfma_frac`.
    kernels = {'conv_igemm': ('conv_igemm_kernel (implicit GEMM, v_mfma_f32_32x32x2_f32)', 1.0),
               'conv_wino': ('conv_wino_kernel (fused Winograd F(2x2,3x3), v_mfma_f32_16x16x4_f32)', 2.25)}
    fams = {k: summ[k] for k in kernels if k in summ}
    dom = max(fams, key=lambda k: fams[k][1])
    n_conv, conv_ms, conv_flops = fams[dom]
    achieved = conv_flops / (conv_ms * 1e-3) / 1e12
    for k, (n_k, ms_k, fl_k) in fams.items():
        stages[k]['algorithmic_TFLOPs'] = round(fl_k / (ms_k * 1e-3) / 1e12, 2)
        stages[k]['mfma_frac'] = round(fl_k / kernels[k][1] / (ms_k * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4)
    # HBM/fabric bytes per launch from the PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE on this same command,
    # corrected as MI355X_MICROARCH.md prescribes: FETCH_SIZE x2 for 16-byte-per-lane loads, WRITE_SIZE exact);
    # measured offline because counters need the profiler, committed under profiles/
    traffic = None
    pmc_path = os.path.join(ROOT, 'profiles', 'r01_pmc_traffic_b32_640.json')
    if args.batch == 32 and args.size == 640 and os.path.exists(pmc_path):
        pmc = json.load(open(pmc_path)).get(dom)
        if pmc:
            traffic = round(pmc['hbm_read_bytes_per_launch_x2corr'] + pmc['hbm_write_bytes_per_launch'])
    alg_bytes = round(timer.bytes.get(dom, 0.0) / n_conv) if n_conv else None

    total_images = world * args.batch * args.steps
    out = {
        'metric': f'images/sec at batch {args.batch}, {args.size}x{args.size}, YOLOv3-80 (backbone -> FPN -> head -> decode -> NMS)',
        'value': round(total_images / elapsed, 2),
        'unit': 'images/sec',
        'n_gpus': world,
        'steps': args.steps,
        'warmup': args.warmup,
        'ms_per_step': round(elapsed / args.steps * 1e3, 3),
        'higher_is_better': True,
        'scaling': 'weak',
        'vs_baseline': None,
        'dtype': 'f32',
        'data': 'synthetic',
        'config': {'workload': f'yolov3_80 (Darknet-53 + YOLOv3 FPN/head + decode + conf 0.005/top-512/NMS 0.45), '
                               f'batch {args.batch}/GPU, {args.size}x{args.size}, random-init calibrated weights',
                   'global_batch': world * args.batch, 'image_size': args.size, 'parallelism': f'dp{world}',
                   'exchange': 'all-gather of 14 340 B detection records' if world > 1 else 'none'},
        'roofline': {'bound': 'mfma', 'kernel': kernels[dom][0],
                     'achieved': round(achieved, 2), 'peak': PEAK_FP32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                     'frac': round(achieved / PEAK_FP32_MFMA_TFLOPS, 4),
                     'mfma_frac': round(achieved / kernels[dom][1] / PEAK_FP32_MFMA_TFLOPS, 4),
                     'note': 'achieved = algorithmic (direct-form) FLOPs / HIP-event time; mfma_frac = multiplies actually '
                             'issued on the matrix pipe / peak (Winograd issues 1/2.25 of the algorithmic count)',
                     'traffic': traffic,
                     'traffic_unit': 'bytes per launch (PMC, profiles/r01_pmc_traffic_b32_640.json)',
                     'algorithmic_bytes_per_launch': alg_bytes,
                     'launches_per_step': n_conv / args.steps,
                     'avg_launch_ms': round(conv_ms / n_conv, 4),
                     'algorithmic_gflop_per_launch': round(conv_flops / n_conv / 1e9, 3)},
        'stages': stages,
        'nms_p50_ms': stages['postprocess']['p50_ms'],
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out['cpu_baseline'] = cpu_baseline(cfg, args.size, args.cpu_sample_batch, args.cpu_repeats, args.cpu_threads)
        out['cpu_baseline']['gpu_over_cpu'] = round(out['value'] / out['cpu_baseline']['value'], 1)
    if rank == 0:
        print(json.dumps(out))
    if world > 1 or rehearse:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
