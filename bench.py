"""Benchmark of the detection inference hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1: starts its own N ranks, one per GPU, over RCCL)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...      (the driver's launcher: same ranks)

One step = one pass of the hot path over one batch resident in HBM:
    images [32,3,640,640] -> Darknet-53 -> YOLOv3 FPN -> head -> decode -> conf filter (0.005)
    -> top-512 -> class-aware NMS (0.45) [-> ONE all-gather of the fixed-size detection records when N > 1]
Workload = BASELINE.json configs[1] (yolov3_80, batch 32 per GPU, 640x640, synthetic weights/images); `--config
efficientdet-d1 --batch 16` and `--config d1_fcs2_atss` are configs[2] and [3].  ONE global batch of N x batch
images (image i is a pure function of i) is cut into contiguous shards (parallel.shard_range), so `--verify` can
compare the gathered records with a 1-GPU pass over the same images.
Metric: images/sec over all GPUs (weak scaling: `--batch` images per GPU).

The JSON line also carries
  roofline     -- the dominant kernel family (most time per step).  YOLOv3: a conv family on FP32 MFMA --
                  `frac` = multiplies actually issued on the matrix pipe / the 157.3 TFLOP/s FP32-MFMA peak;
                  `algorithmic_frac` = direct-form FLOPs / peak (exceeds `frac` by 2.25x for the Winograd kernel,
                  which removes multiplies instead of executing them).  EfficientDet family: HBM-bound --
                  `frac` = algorithmic bytes / HIP-event time / 8 TB/s.  Times are HIP events on the launch
                  stream inside the timed region; `traffic` = PMC bytes per launch from a committed rocprofv3
                  pass (profiles/), tagged with the file it came from, or null.
  cpu_baseline -- the CPU oracle (port of the reference path) timed on this host's cores on a bounded sample
                  (rank 0, N=1 only): forward and post-process separately, median of 5
  stages       -- per-kernel-family time per step, incl. the NMS launch (latency-bound; p50 reported)
  parity_check -- the timed step's own records against the oracle (first images of the batch): NMS decisions exact, and --
                  with the CPU baseline's oracle forward at hand -- scores / boxes within 1e-4 and the detection sets equal
                  wherever no decision sits inside the observed round-off; a failure exits with code 4
`--nms-worst` measures the isolated NMS worst cases of SURVEY 8d instead (512 boxes in 1 / in 80 classes).
`--profile` is the command to put behind `rocprofv3 ... --`: nothing but warm-up and timed passes is launched.
"""
import argparse
import json
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3
PEAK_BF16_MFMA_TFLOPS = 2516.6      # dense: 256 CUs x 4 096 FLOP/clk (v_mfma_f32_32x32x16_bf16: 32 768 FLOP / 32 cycles / SIMD) x 2.4 GHz
PEAK_HBM_GBS = 8000.0
DEFAULT_BATCH = {'yolov3_80': 32, 'efficientdet-d1': 16, 'd1_fcs2_atss': 32}
WORKLOADS = {
    'yolov3_80': 'yolov3_80 (Darknet-53 + YOLOv3 FPN/head + decode + conf 0.005/top-512/NMS 0.45)',
    'efficientdet-d1': 'efficientdet-d1 (EfficientNet-B1 + 4x BiFPN + EfDetHead + RetinaNet decode + conf 0.005/top-512/NMS 0.5)',
    'd1_fcs2_atss': 'd1_fcs2_atss (EfficientNet-B1 + 4x BiFPN + EfDetHead + FCOS decode + conf 0.005/top-512/NMS 0.5)',
}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=None, help='images per GPU (default: the BASELINE batch of --config)')
    ap.add_argument('--size', type=int, default=640)
    ap.add_argument('--config', default='yolov3_80', help='yolov3_80 (headline) | efficientdet-d1 | d1_fcs2_atss')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--graph', action='store_true', help='(default) replay the step from a captured hipGraph')
    ap.add_argument('--eager', action='store_true', help='issue every launch of the step from the host instead of replaying a hipGraph')
    ap.add_argument('--verify', action='store_true',
                    help='after the timed region: the gathered records must equal a 1-GPU pass over the same global batch')
    ap.add_argument('--nms-worst', action='store_true', help='isolated NMS worst cases (512 boxes, 1 and 80 classes)')
    ap.add_argument('--cpu-sample-batch', type=int, default=2)
    ap.add_argument('--cpu-repeats', type=int, default=5)
    ap.add_argument('--cpu-threads', type=int, default=16)
    ap.add_argument('--parity-images', type=int, default=6,
                    help='images of the timed batch the parity check compares with the CPU oracle forward (the CPU baseline sample '
                         'covers the first --cpu-sample-batch of them; the rest cost one more oracle forward each)')
    ap.add_argument('--dry-run-launch', action='store_true', help='with --gpus N > 1: print the launcher command as JSON and exit')
    ap.add_argument('--no-other-configs', action='store_true',
                    help='headline run only: skip the short runs of the other single-GPU BASELINE configs (`other_configs`)')
    ap.add_argument('--no-power-probe', action='store_true', help='skip the 1.5 s of extra replays under rocm-smi sampling (board power, shader clock)')
    ap.add_argument('--lanes', default=None,
                    help="batch lanes of the replayed graph: a number, 'auto' (capture with 1 and 2, keep the faster: rank 0 decides "
                         "for all ranks), default: MYDET_LANES or the model's batch_lanes_hint -- the rule api.Detector uses")
    ap.add_argument('--profile', action='store_true',
                    help='for rocprofv3: only capture warm-up + warm-up + timed replays run (no lane trial, no pricing pass, no NMS '
                         'p50 loop, no CPU baseline): every kernel count in the trace = launches per step x `passes` of the JSON line')
    return ap.parse_args(argv)


def launch_command(args, port, argv=None):
    """The command `--gpus N` starts when no launcher did: one rank per GPU of this node over RCCL, rendezvous on 127.0.0.1
    (the driver's own launcher line)."""
    return [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
            '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + list(sys.argv[1:] if argv is None else argv)


def self_launch(args):
    """--gpus N > 1 without a launcher: start N ranks (one process per GPU) BEFORE this process touches the GPU and
    exit with their return code.  torch.cuda.device_count() does not initialise HIP; nothing else here does either."""
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    cmd = launch_command(args, port)
    if args.dry_run_launch:
        print(json.dumps({'launch': cmd}))
        return 0
    import torch
    have = torch.cuda.device_count()
    if have < args.gpus:
        sys.stderr.write(f'bench.py: --gpus {args.gpus} but this machine exposes {have} GPU(s); nothing was run\n')
        return 2
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    return subprocess.run(cmd, env=env).returncode


def cpu_model_name():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def cpu_baseline(config, cfg, size, sample_batch, repeats, max_threads=16):
    """Oracle forward + post-process on the host cores (reported baseline, not the target): median of `repeats`
    after one warm-up, forward and per-image post_process timed separately."""
    import torch
    from mydetection_amd import synth
    from mydetection_amd.models.general import state_dict_template
    from oracle import postprocess as opp
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        pass
    cores = min(cores, max_threads)         # a 1-GPU box owns a 16-CPU share of the host, not all of it
    torch.set_num_threads(cores)
    sd = synth.make_state_dict(state_dict_template(config), config)
    x = synth.make_image_set(0, sample_batch, size, cfg['general.input_format'])
    if config == 'yolov3_80':
        from oracle import yolov3 as oy
        fwd = lambda: oy.forward(x, sd)                                   # noqa: E731
    else:
        from oracle import efficientdet as oe
        fwd = lambda: oe.forward(x, sd, config)                           # noqa: E731
    conf, nms = cfg['test.ap_conf_thres'], cfg['test.nms_thres']

    def once():
        t0 = time.perf_counter()
        with torch.no_grad():
            bb, ci, sc = fwd()
        t1 = time.perf_counter()
        for b in range(sample_batch):
            opp.post_process(bb[b].numpy(), ci[b].numpy(), sc[b].numpy(), conf, nms)
        return t1 - t0, time.perf_counter() - t1
    once()
    runs = [once() for _ in range(repeats)]
    with torch.no_grad():
        bb, ci, sc = fwd()                           # the oracle's candidates of the sample images: bench's parity check
    f_med = statistics.median(r[0] for r in runs)
    p_med = statistics.median(r[1] for r in runs)
    tot = statistics.median(r[0] + r[1] for r in runs)
    return {'value': round(sample_batch / tot, 3), 'unit': 'images/sec', 'cores': cores, 'kind': 'port',
            'cpu': cpu_model_name(),
            'forward_ms_per_image': round(f_med / sample_batch * 1e3, 2),
            'post_process_ms_per_image': round(p_med / sample_batch * 1e3, 3),
            'sample': f'oracle/ (torch-CPU restatement of the reference path + C NMS), {config} batch {sample_batch} '
                      f'{size}x{size}, forward and per-image post_process timed separately, median of {repeats} after 1 warm-up'}, \
        (bb.numpy(), ci.numpy(), sc.numpy())


def oracle_candidates(config, cfg, size, lo, hi, max_threads=16):
    """The CPU oracle's candidates (bbox, class_idx, score as numpy) of images [lo, hi) of the global synthetic batch:
    the checker side of `parity_check` (never timed as the product)."""
    import numpy as np
    import torch
    from mydetection_amd import synth
    from mydetection_amd.models.general import state_dict_template
    if hi <= lo:
        return None
    torch.set_num_threads(min(len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1), max_threads))
    sd = synth.make_state_dict(state_dict_template(config), config)
    x = synth.make_image_set(lo, hi, size, cfg['general.input_format'])
    with torch.no_grad():
        if config == 'yolov3_80':
            from oracle import yolov3 as oy
            bb, ci, sc = oy.forward(x, sd)
        else:
            from oracle import efficientdet as oe
            bb, ci, sc = oe.forward(x, sd, config)
    return bb.numpy(), ci.numpy(), sc.numpy()


def parity_check(cand, rec, conf, nms, oracle_cand=None, images=2, extra_conf=()):
    """SURVEY 8d "parity gates run with every measurement": the first `images` images of the timed batch.
    * always: the GPU's detection records == the oracle's post_process (oracle/postprocess.py: reference filter / top-512
      / class-aware NMS order) of the GPU's own candidates -- count, candidate indices, classes, order, exactly;
    * with the oracle's forward at hand (the cpu_baseline leg computes it for its timing anyway): every candidate's score
      within 1e-4 and box within 1e-4 + 1e-4 |ref| (north_star), and the detection sets equal to the oracle's own
      wherever no post-processing decision sits within twice the observed score error of flipping
      (oracle.postprocess.decision_margins)."""
    import numpy as np
    from oracle import postprocess as opp
    bb, ci, sc = (t[:images].cpu().numpy() for t in cand)
    n = bb.shape[0]
    count = rec['count'][:n].cpu().numpy()
    index = rec['index'][:n].cpu().numpy()
    cls = rec['class_idx'][:n].cpu().numpy()
    out = {'images': int(n), 'nms_equals_oracle_on_gpu_candidates': True}
    for b in range(n):
        _, oc, _, oi = opp.post_process(bb[b], ci[b], sc[b], conf, nms)
        k = int(count[b])
        if k != len(oi) or not np.array_equal(index[b, :k], oi) or not np.array_equal(cls[b, :k], oc):
            out['nms_equals_oracle_on_gpu_candidates'] = False
    ok = out['nms_equals_oracle_on_gpu_candidates']
    if oracle_cand is not None:
        obb, oci, osc = (a[:n] for a in oracle_cand)
        s_err = float(np.abs(sc - osc).max())
        b_err = float((np.abs(bb - obb) / (1e-4 + 1e-4 * np.abs(obb))).max())
        safe = equal = 0
        kept = {f'conf_{conf}': [int(k) for k in count]}     # detections per image at every threshold: the sets are not empty
        jac = []                                             # agreement of the kept candidate ids with the oracle's own detections,
                                                             # every (image, threshold) pair, margin-safe or not (Jaccard index)

        def jaccard(mine, ref):
            u = len(np.union1d(mine, ref))
            return 1.0 if u == 0 else len(np.intersect1d(mine, ref)) / u
        for b in range(n):
            _, rc, _, ri = opp.post_process(obb[b], oci[b], osc[b], conf, nms)
            k = int(count[b])
            jac.append(jaccard(index[b, :k], ri))
            if opp.decision_margins(osc[b], oci[b], conf, eps=max(2.0 * s_err, 2e-6)) is None:
                safe += 1
                equal += int(k == len(ri) and np.array_equal(index[b, :k], ri) and np.array_equal(cls[b, :k], rc))
        extra = {f'conf_{conf}': f'{equal}/{safe}'}
        # the same comparison at the other two thresholds (fresh post-process launches on the step's candidates)
        from mydetection_amd.utils.structures import batched_post_process
        for t in extra_conf:
            r_t = batched_post_process(*(c[:n] for c in cand), t, nms)
            cnt_t, idx_t, cls_t = (r_t[k].cpu().numpy() for k in ('count', 'index', 'class_idx'))
            kept[f'conf_{t}'] = [int(k) for k in cnt_t]
            s_t = e_t = 0
            for b in range(n):
                _, rc, _, ri = opp.post_process(obb[b], oci[b], osc[b], t, nms)
                k = int(cnt_t[b])
                jac.append(jaccard(idx_t[b, :k], ri))
                if opp.decision_margins(osc[b], oci[b], t, eps=max(2.0 * s_err, 2e-6)) is None:
                    s_t += 1
                    e_t += int(k == len(ri) and np.array_equal(idx_t[b, :k], ri) and np.array_equal(cls_t[b, :k], rc))
            extra[f'conf_{t}'] = f'{e_t}/{s_t}'
            safe, equal = safe + s_t, equal + e_t
        out['sets_equal_by_threshold'] = extra
        out['detections_per_image'] = kept
        out.update({'max_score_err': s_err, 'max_box_err_over_tol': round(b_err, 4),
                    'class_id_agreement': round(float((ci == oci).mean()), 6),
                    'sets_equal': f'{equal}/{safe} margin-safe (image, threshold) pairs equal the oracle\'s detections (count, candidate indices, classes, order)',
                    'margin_safe_pairs': safe, 'tolerance': 'scores 1e-4 abs; boxes 1e-4 + 1e-4 |ref|',
                    'kept_ids_jaccard': {'pairs': len(jac), 'min': round(min(jac), 4), 'mean': round(sum(jac) / len(jac), 4),
                                         'note': 'kept candidate ids vs the oracle\'s own detections over ALL (image, threshold) pairs; '
                                                 'below 1 only through decisions inside the float32 round-off band'}})
        ok = ok and s_err <= 1e-4 and b_err <= 1.0 and equal == safe and min(jac) >= 0.95
    out['ok'] = bool(ok)
    return out


def nms_worst_cases(dev, iters=200, warm=20):
    """exactly 512 boxes per image in 1 class and spread over 80 classes, B = 1 and B = 32; p50 / p95 of
    the post-process launch over `iters` stream-synchronised iterations (HIP events around each launch)."""
    import numpy as np
    import torch
    from mydetection_amd import ops
    out = {}
    rng = np.random.Generator(np.random.PCG64(5))
    for n_cls in (1, 80):
        for B in (1, 32):
            cx = rng.random((B, 512), dtype=np.float32) * 600 + 20
            cy = rng.random((B, 512), dtype=np.float32) * 600 + 20
            wh = rng.random((B, 512, 2), dtype=np.float32) * 120 + 20
            bb = torch.from_numpy(np.concatenate([cx[..., None], cy[..., None], wh], axis=-1)).to(dev)
            ci = torch.from_numpy(rng.integers(0, n_cls, size=(B, 512)).astype(np.int64)).to(dev)
            sc = torch.from_numpy((rng.random((B, 512), dtype=np.float32) * 0.9 + 0.05)).to(dev)
            for _ in range(warm):
                rec = ops.postprocess(bb, ci, sc, 0.005, 0.45)
            torch.cuda.synchronize()
            spans = []
            for _ in range(iters):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                rec = ops.postprocess(bb, ci, sc, 0.005, 0.45)
                e1.record()
                torch.cuda.synchronize()
                spans.append(e0.elapsed_time(e1))
            spans.sort()
            out[f'{n_cls}_class_B{B}'] = {'p50_ms': round(spans[len(spans) // 2], 4), 'p95_ms': round(spans[int(len(spans) * 0.95)], 4),
                                          'kept_per_image': round(float(rec['count'].float().mean()), 1), 'iterations': iters}
    return out


def main():
    args = parse_args()
    args.graph = not args.eager          # the product's default path (api.Detector replays captured graphs, MYDET_GRAPH=1)
    if args.gpus > 1 and 'RANK' not in os.environ:
        sys.exit(self_launch(args))

    import torch
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        sys.exit(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}')
    assert torch.cuda.is_available(), 'bench.py needs MI355X GPUs'
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    rehearse = world == 1 and 'RANK' in os.environ and os.environ.get('MYDET_REHEARSE_RCCL') == '1'
    dist_on = world > 1 or rehearse       # (rehearse: one-rank RCCL group, exercises the collective path on a 1-GPU box)
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', device_id=dev)          # nccl == RCCL on ROCm

    from mydetection_amd import _lib, ops, parallel, synth
    from mydetection_amd.models.general import name_to_model
    from mydetection_amd.utils.structures import batched_post_process
    _lib.lib()                                                   # loud failure if the HIP library is missing

    if args.nms_worst:
        print(json.dumps({'metric': 'NMS worst-case latency (filter + top-512 + class-aware NMS 0.45), 512 boxes per image',
                          'unit': 'ms', 'cases': nms_worst_cases(dev)}))
        return

    ctx = {'dev': dev, 'world': world, 'rank': rank, 'dist_on': dist_on, 'rehearse': rehearse}
    out, code = measure(args, ctx)
    # the other single-GPU BASELINE configs, briefly, inside the driver's default command (the headline fields above stay as
    # they are): configs[2], configs[3] and the 512 x 512 batch north_star asks for -- value, roofline and parity gate each
    if (out is not None and world == 1 and not dist_on and args.config == 'yolov3_80' and args.size == 640 and args.batch is None
            and not args.eager and not args.no_other_configs and not args.verify):
        others = {}
        for name, key, bsz, size in OTHER_CONFIGS:
            a = argparse.Namespace(**vars(args))
            a.config, a.batch, a.size = name, bsz, size
            a.steps, a.warmup = min(args.steps, 10), min(args.warmup, 3)
            a.no_cpu_baseline, a.parity_images, a.lanes, a.no_power_probe = True, 2, None, True
            t0 = time.perf_counter()
            o, c = measure(a, ctx)
            code = code or c
            others[key] = brief_line(o, time.perf_counter() - t0)
        out['other_configs'] = others
        # the headline once more with every direct conv on the float32 matrix instruction (MYDET_CONV_SPLIT_BF16=0's path): what
        # the split-bf16 kernels contribute, on this box, in this call
        from mydetection_amd import ops
        if ops.SPLIT_BF16:
            a = argparse.Namespace(**vars(args))
            a.steps, a.warmup, a.no_cpu_baseline, a.parity_images, a.no_power_probe = min(args.steps, 10), min(args.warmup, 3), True, 2, True
            t0 = time.perf_counter()
            ops.SPLIT_BF16 = False
            try:
                o, c = measure(a, ctx)
            finally:
                ops.SPLIT_BF16 = True
            code = code or c
            out['float32_mfma_only'] = {'value': o['value'], 'unit': o['unit'], 'ms_per_step': o['ms_per_step'], 'steps': o['steps'],
                                        'max_score_err': o['parity_check'].get('max_score_err'), 'parity_ok': o['parity_check'].get('ok'),
                                        'note': 'the same step with every direct conv on v_mfma_f32_32x32x2_f32 (no split-bf16 kernels)',
                                        'wall_s': round(time.perf_counter() - t0, 1)}
    if out is not None:
        out['summary'] = summary_of(out)          # LAST key, < 1 500 characters: what a 2 000-character tail of this line still shows
        print(json.dumps(out))
    if dist_on:
        torch.distributed.destroy_process_group()
    if code:
        sys.exit(code)


OTHER_CONFIGS = (('efficientdet-d1', 'efficientdet-d1_b16_640', 16, 640),        # BASELINE configs[2]
                 ('d1_fcs2_atss', 'd1_fcs2_atss_b32_640', 32, 640),             # BASELINE configs[3]
                 ('yolov3_80', 'yolov3_80_b32_512', 32, 512))                   # north_star: 512 x 512 batches as well


def summary_of(out):
    """Compact digest of the whole line (the driver's record keeps the tail of stdout only).  Per other config:
    [images/s, ms_per_step, roofline frac, fused_min_frac, launches_per_lane, parity ok, margin_safe_pairs, detections per image at
    the three thresholds of image 0]."""
    def pc(o):
        p_ = o.get('parity_check') or {}
        d = p_.get('detections_per_image') or {}
        return [p_.get('ok'), p_.get('margin_safe_pairs'), [v[0] for v in d.values() if v]]
    s = {'value': out['value'], 'ms': out['ms_per_step'], 'frac': out['roofline'].get('frac'),
         'traffic_x': (round(out['roofline']['traffic'] / out['roofline']['algorithmic_bytes_per_launch'], 2)
                       if out['roofline'].get('traffic') and out['roofline'].get('algorithmic_bytes_per_launch') else None),
         'launches_per_lane': out.get('launches_per_lane'), 'parity': pc(out),
         'decode_frac_hbm': (out['stages'].get('decode') or {}).get('frac_hbm_peak'), 'nms_p50_ms': out.get('nms_p50_ms'),
         'cpu_img_s': (out.get('cpu_baseline') or {}).get('value'), 'cpu_cores': (out.get('cpu_baseline') or {}).get('cores')}
    if out.get('power'):
        s['board_w'], s['sclk_mhz'] = out['power'].get('board_w'), out['power'].get('sclk_mhz')
    if out.get('float32_mfma_only'):
        s['float32_mfma_only'] = out['float32_mfma_only']['value']
    if out.get('rccl_ranks') is not None:
        s['rccl_ranks'], s['exchange_ms'] = out['rccl_ranks'], out.get('exchange_ms_per_step')
    for k, o in (out.get('other_configs') or {}).items():
        r = o['roofline']
        s[k] = [o['value'], o['ms_per_step'], r.get('frac'), r.get('fused_min_frac'), o['launches_per_lane']] + pc(o)
    return s


def brief_line(o, wall_s):
    """The fields of a full line that `other_configs` carries."""
    keep = ('bound', 'kernel', 'achieved', 'peak', 'unit', 'frac', 'hbm_frac', 'mfma_frac', 'step_frac', 'fused_min_frac',
            'fused_min_bytes', 'step_algorithmic_bytes', 'launches_per_step', 'avg_launch_ms', 'traffic', 'traffic_source',
            'algorithmic_bytes_per_launch')
    return {'metric': o['metric'], 'value': o['value'], 'unit': o['unit'], 'ms_per_step': o['ms_per_step'], 'steps': o['steps'],
            'warmup': o['warmup'], 'kernel_ms_per_step': o['kernel_ms_per_step'], 'batch_lanes': o['config']['batch_lanes'],
            'launches_per_lane': o['launches_per_lane'], 'workload': o['config']['workload'], 'nms_p50_ms': o['nms_p50_ms'],
            'roofline': {k: o['roofline'][k] for k in keep if k in o['roofline']},
            'parity_check': o['parity_check'], 'wall_s': round(wall_s, 1)}


def power_probe(step, seconds=3.0):
    """Board power and shader clock (rocm-smi, a child process, sampled while `step` keeps replaying for `seconds`) -- or None when
    rocm-smi is not there.  Not part of the timed region.  The headline step runs at the board's power cap: its
    time is its energy over the cap, which this pair of numbers shows next to the measurement."""
    import re
    import shutil
    import subprocess
    import threading
    import torch
    if shutil.which('rocm-smi') is None:
        return None
    samples, stop = [], []

    def sampler():
        while not stop:
            try:
                txt = subprocess.run(['rocm-smi', '--showpower', '--showclocks', '--showmaxpower'], capture_output=True, text=True, timeout=10).stdout
            except Exception:
                return
            pw = re.search(r'Current Socket Graphics Package Power \(W\): ([\d.]+)', txt) or re.search(r'Average Graphics Package Power \(W\): ([\d.]+)', txt)
            ck = re.search(r'sclk clock level: \d+: \((\d+)Mhz\)', txt)
            cap = re.search(r'Max Graphics Package Power \(W\): ([\d.]+)', txt)
            if pw and ck:
                samples.append((float(pw.group(1)), int(ck.group(1)), float(cap.group(1)) if cap else None))
    th = threading.Thread(target=sampler, daemon=True)
    t_end = time.perf_counter() + seconds
    th.start()
    n = 0
    while time.perf_counter() < t_end:
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        n += 5
    stop.append(1)
    th.join(15)
    if not samples:
        return None
    tail = samples[len(samples) // 2:]                                # rocm-smi reports a running average: the second half has settled
    return {'board_w': round(sum(a for a, _, _ in tail) / len(tail), 1), 'board_w_max': max(a for a, _, _ in samples), 'cap_w': samples[-1][2],
            'sclk_mhz': round(sum(b for _, b, _ in tail) / len(tail)), 'samples': len(samples), 'steps_replayed': n,
            'note': 'rocm-smi while the step keeps replaying after the timed region (second half of the samples); near cap_w = the step is '
                    'power-bound: its time is its energy over the cap (DESIGN.md section 5; a 14 s run reads 1 366-1 374 W on the headline)'}


def measure(args, ctx):
    """One configuration measured as the module docstring says: returns (the JSON line as a dict or None on ranks > 0 and in
    --profile mode, exit code: 0, 3 = --verify failed, 4 = the parity check failed)."""
    import torch
    dev, world, rank, dist_on, rehearse = ctx['dev'], ctx['world'], ctx['rank'], ctx['dist_on'], ctx['rehearse']
    from mydetection_amd import _lib, ops, parallel, synth
    from mydetection_amd.models.general import name_to_model
    from mydetection_amd.utils.structures import batched_post_process
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        model, cfg = name_to_model(args.config)
    model.load_state_dict(synth.make_state_dict(model.state_dict(), args.config), strict=True)
    model = model.eval().to(dev)
    conf, nms = cfg['test.ap_conf_thres'], cfg['test.nms_thres']
    batch = args.batch or DEFAULT_BATCH.get(args.config, 32)
    total = world * batch
    # ONE global batch, sharded contiguously; this rank's shard is resident in HBM before the timed region
    lo, hi = parallel.shard_range(total, rank, world)
    x = synth.make_image_set(lo, hi, args.size, cfg['general.input_format']).to(dev)

    graphed = None
    lanes_rule = None
    if args.graph:
        from mydetection_amd.graph import GraphedPath
        # lane count: ONE rule for every rank (lanes change the last float bits, and --verify compares ranks bit for bit):
        # --lanes / MYDET_LANES when numeric, else the model's hint (what api.Detector does); 'auto' = rank 0 times both
        # captures and tells the others
        want = args.lanes if args.lanes is not None else os.environ.get('MYDET_LANES', '')
        if str(want) == 'auto' and not args.profile:
            lanes_rule = 'auto (rank 0 timed 1 and 2 lanes' + (', broadcast)' if dist_on else ')')
            if rank == 0:
                graphed = GraphedPath(model, x, conf, nms, lanes='auto')
            n_l, lanes_per_rank = parallel.agree_on_lanes(graphed.lanes if rank == 0 else None, device=dev)
            if rank != 0:
                graphed = GraphedPath(model, x, conf, nms, lanes=n_l)
        else:
            n_l = int(want) if str(want).isdigit() else int(getattr(model, 'batch_lanes_hint', 1))
            if n_l < 1 or batch % max(n_l, 1):
                n_l = 1
            lanes_rule = ('--lanes' if args.lanes is not None and str(want).isdigit() else
                          'MYDET_LANES' if str(want).isdigit() else 'model.batch_lanes_hint')
            n_l, lanes_per_rank = parallel.agree_on_lanes(n_l, device=dev)     # every rank applied the rule itself: a differing rank raises
            graphed = GraphedPath(model, x, conf, nms, lanes=n_l)
        assert graphed.lanes == n_l

    def local_records(inp=None, pricing=False):
        if graphed is not None and inp is None:
            return graphed()
        if graphed is not None and not pricing:
            return {k: v.clone() for k, v in graphed.eager(inp).items()}      # the replayed decomposition, host-issued
        with torch.no_grad():
            src = x if inp is None else inp
            if pricing and graphed is not None and graphed.lanes > 1:
                # the kernels the replayed graph runs (a lane sees batch / lanes images: other tile rules than the full
                # batch), lane after lane on ONE stream so that their event times do not overlap
                for part in src.tensor_split(graphed.lanes):
                    bb, ci, sc = model.forward_candidates(part)
                    out = batched_post_process(bb, ci, sc, conf, nms)
                return out
            bb, ci, sc = model.forward_candidates(src)
            return batched_post_process(bb, ci, sc, conf, nms)

    exchange = []                          # (start, end) HIP events around the all-gather of every step, on the launch stream

    def step():
        local = local_records()
        if not dist_on:
            return local
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = parallel.gather_detections(local, always=rehearse, total=total)
        e1.record()                        # the launch stream has waited for RCCL's stream here (blocking collective semantics)
        exchange.append((e0, e1))
        return out

    def barrier():
        if dist_on:
            torch.distributed.barrier()

    for _ in range(args.warmup):
        rec = step()
    torch.cuda.synchronize()

    if graphed is None:
        ops.TIMER = ops.KernelTimer(chain=True)                  # HIP events on the launch stream; one event per launch boundary
    exchange.clear()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        if ops.TIMER is not None:
            ops.TIMER.cut()                                      # the exchange between steps is not a timed launch
        rec = step()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    timer, ops.TIMER = ops.TIMER, None
    power = None
    if rank == 0 and world == 1 and not dist_on and not args.profile and not getattr(args, 'no_power_probe', False):
        power = power_probe(step)

    # launches of the step's kernels over the whole process: GraphedPath runs two eager warm-up passes before a capture
    passes = (None if lanes_rule and lanes_rule.startswith('auto') else
              (2 if graphed is not None else 0) + args.warmup + args.steps)
    if args.profile:
        # nothing else may launch: the profiler's per-kernel counts are then launches per step x `passes` (graph mode: two
        # eager warm-up passes inside GraphedPath before the capture, which itself launches nothing)
        if dist_on:
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            elapsed = float(t.item())
        if rank == 0:
            print(json.dumps({'metric': f'images/sec at batch {batch}, {args.size}x{args.size}, {args.config} (profile run)',
                              'value': round(total * args.steps / elapsed, 2), 'unit': 'images/sec', 'n_gpus': world,
                              'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(elapsed / args.steps * 1e3, 3),
                              'passes': passes, 'batch_lanes': graphed.lanes if graphed is not None else 1,
                              'mode': 'hipGraph replay' if graphed is not None else 'eager',
                              'note': 'profile run: every kernel of the step appears launches-per-step x passes times in the trace'}))
        return None, 0

    # the parity check's GPU side: the candidates and records of the LAST TIMED STEP (first images of this rank's shard)
    n_pc = max(1, min(args.parity_images, batch))
    with torch.no_grad():
        if graphed is not None:
            pc_cand = tuple(t[:n_pc].clone() for t in graphed.cand)
            pc_rec = {k: rec[k][lo:lo + n_pc].clone() for k in ('count', 'index', 'class_idx')}
        else:
            cand_e = model.forward_candidates(x)
            rec_e = batched_post_process(*cand_e, conf, nms)
            pc_cand = tuple(t[:n_pc].clone() for t in cand_e)
            pc_rec = {k: rec_e[k][:n_pc].clone() for k in ('count', 'index', 'class_idx')}
            del cand_e, rec_e

    if timer is None:        # graph replay hides the launches from the event timer: price the kernels eagerly, after
        ops.TIMER = ops.KernelTimer()
        for _ in range(args.steps):
            local_records(x, pricing=True)
        torch.cuda.synchronize()
        timer, ops.TIMER = ops.TIMER, None

    # NMS latency proper: the post-process launch alone on this step's candidates, its own start / stop events per launch
    # (the chained timer of the timed region charges any inter-launch gap to the next kernel)
    with torch.no_grad():
        cand = model.forward_candidates(x)
    nms_spans = []
    for i in range(120):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        batched_post_process(*cand, conf, nms)
        e1.record()
        if i >= 20:
            nms_spans.append((e0, e1))
    torch.cuda.synchronize()
    nms_spans = sorted(a.elapsed_time(b) for a, b in nms_spans)
    del cand

    if dist_on:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())

    verify = None
    if args.verify:
        # rank 0 runs every shard's images through its own GPU, in the shard composition the ranks used, and the
        # gathered records must equal that bit for bit (count, candidate indices, classes, scores, boxes)
        ok, checked = True, 0
        if rank == 0:
            got = {k: v.clone() for k, v in rec.items() if k != 'records'}
            for r in range(world):
                rlo, rhi = parallel.shard_range(total, r, world)
                xr = x if r == 0 else synth.make_image_set(rlo, rhi, args.size, cfg['general.input_format']).to(dev)
                ref = local_records(xr)
                for k in ('count', 'index', 'class_idx', 'score', 'bbox'):
                    ok = ok and torch.equal(got[k][rlo:rhi], ref[k])
                checked += rhi - rlo
            ok = ok and got['count'].shape[0] == total and int(got['count'].sum()) > 0
        verify = {'ok': bool(ok), 'images': checked, 'against': '1-GPU pass over the same global batch (rank 0)'}

    summ = timer.summary()
    stages = {k: {'launches_per_step': v[0] / args.steps, 'ms_per_step': round(v[1] / args.steps, 4)} for k, v in summ.items()}
    kernel_ms = sum(v[1] for v in summ.values()) / args.steps
    pp_spans = sorted(a.elapsed_time(b) for a, b, _ in timer.spans['postprocess'])
    stages['postprocess']['in_step_p50_ms'] = round(statistics.median(pp_spans), 4)
    stages['postprocess']['p50_ms'] = round(statistics.median(nms_spans), 4)
    stages['postprocess']['p95_ms'] = round(nms_spans[min(len(nms_spans) - 1, int(len(nms_spans) * 0.95))], 4)
    stages['postprocess']['p50_note'] = ('p50/p95: 100 back-to-back launches on the step\'s candidates after the timed region, own start/stop '
                                         'events per launch; in_step_p50_ms: the launch inside the timed steps (chained events)')
    stages['postprocess']['mean_detections_per_image'] = round(float(rec['count'].float().mean()), 1)
    if 'decode' in summ:
        dec_ms, dec_bytes = summ['decode'][1], summ['decode'][2]
        stages['decode']['achieved_GBs'] = round(dec_bytes / (dec_ms * 1e-3) / 1e9, 1)
        stages['decode']['frac_hbm_peak'] = round(dec_bytes / (dec_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)
    if 'sepconv_decode' in summ:
        stages['sepconv_decode']['note'] = ('the towers\' last separable convs with the RetinaNet decode in their epilogue: class logits '
                                            'never reach memory; algorithmic_GBs prices it with the bytes of the three reference layers '
                                            '(depthwise, pointwise, decode)')
    for k, v in summ.items():                                    # algorithmic bytes / time per family
        if timer.bytes.get(k):
            stages[k]['algorithmic_GBs'] = round(timer.bytes[k] / (v[1] * 1e-3) / 1e9, 1)

    if args.config == 'yolov3_80':
        # Dominant kernel = the conv family with the most time per step, priced with the ALGORITHMIC flops of its layers
        # (2*MACs of the direct form).  The Winograd kernels issue 2.25x (F(2x2)) / 4x (F(4x4)) fewer multiplies than that.
        # family -> (kernel, multiplies issued on the matrix pipe per algorithmic multiply, dense peak of the instruction used)
        kernels = {'conv_igemm': ('conv_igemm_kernel (implicit GEMM, v_mfma_f32_32x32x2_f32)', 1.0, PEAK_FP32_MFMA_TFLOPS),
                   'conv_igemm_b3': ('conv_igemm_b3_kernel (the same implicit GEMM with float32-exact split operands: three bfloat16 '
                                     'pieces per operand, six v_mfma_f32_32x32x16_bf16 piece products per k-step, float32 accumulation)',
                                     6.0, PEAK_BF16_MFMA_TFLOPS),
                   'conv_p3': ('conv_p3_kernel (3x3 conv, input patch resident in LDS, split-bf16 operands: six v_mfma_f32_32x32x16_bf16 piece products per k-step)',
                               6.0, PEAK_BF16_MFMA_TFLOPS),
                   'conv_wino': ('conv_wino_kernel (fused Winograd F(2x2,3x3), v_mfma_f32_16x16x4_f32)', 1 / 2.25, PEAK_FP32_MFMA_TFLOPS),
                   'conv_wino4': ('wino4_input_kernel + conv_wino4_kernel (Winograd F(4x4,3x3): input-transform launch + DMA-fed '
                                  'v_mfma_f32_16x16x4_f32 GEMMs with fused output transform, + the K-cut tail\'s piece and fixup '
                                  'launches where the item count leaves a remainder; one layer timed as one unit)', 0.25, PEAK_FP32_MFMA_TFLOPS)}
        fams = {k: summ[k] for k in kernels if k in summ}
        dom = max(fams, key=lambda k: fams[k][1])
        n_k, ms_k, flops_k = fams[dom]
        alg = flops_k / (ms_k * 1e-3) / 1e12
        for k, (n_f, ms_f, fl_f) in fams.items():
            stages[k]['algorithmic_TFLOPs'] = round(fl_f / (ms_f * 1e-3) / 1e12, 2)
            stages[k]['mfma_frac'] = round(fl_f * kernels[k][1] / (ms_f * 1e-3) / 1e12 / kernels[k][2], 4)
            if k in ('conv_igemm_b3', 'conv_p3'):
                stages[k]['mfma_frac_note'] = ('6 x the algorithmic FLOPs issued on the bfloat16 pipe / its 2 516.6 TFLOP/s dense peak; the '
                                               'float32 matrix instruction would need algorithmic_TFLOPs / 157.3 = '
                                               f'{fl_f / (ms_f * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS:.2f} of its peak for this time')
        roofline = {'bound': 'mfma', 'kernel': kernels[dom][0],
                    'achieved': round(alg * kernels[dom][1], 2), 'peak': kernels[dom][2], 'unit': 'TFLOP/s',
                    'frac': round(alg * kernels[dom][1] / kernels[dom][2], 4),
                    'algorithmic_achieved': round(alg, 2),
                    'algorithmic_frac': round(alg / PEAK_FP32_MFMA_TFLOPS, 4),
                    'note': 'achieved/frac = multiplies issued on the matrix pipe (algorithmic direct-form FLOPs x '
                            f'{kernels[dom][1]:.4g}) / HIP-event time vs the dense peak of the instruction used; algorithmic_* = direct-form FLOPs / time vs the FP32-MFMA peak',
                    'algorithmic_gflop_per_launch': round(flops_k / n_k / 1e9, 3)}
    else:
        # the dense / pointwise convs are ONE family here, whichever matrix instruction a layer's launch uses (conv_igemm_kernel on
        # FP32 MFMA, conv_igemm_b3_kernel with split operands on BF16 MFMA): times, FLOPs and bytes added
        merged = 'conv_igemm_b3' in summ and 'conv_igemm' in summ
        if merged:
            a_, b_ = summ['conv_igemm'], summ['conv_igemm_b3']
            summ = dict(summ)
            summ['conv_igemm'] = (a_[0] + b_[0], a_[1] + b_[1], a_[2] + b_[2])
            timer.bytes = dict(timer.bytes)
            timer.bytes['conv_igemm'] = timer.bytes.get('conv_igemm', 0.0) + timer.bytes.get('conv_igemm_b3', 0.0)
        fams = {k: summ[k] for k in summ if timer.bytes.get(k) and not (merged and k == 'conv_igemm_b3')}
        dom = max(fams, key=lambda k: fams[k][1])
        n_k, ms_k, _ = fams[dom]
        gbs = timer.bytes[dom] / (ms_k * 1e-3) / 1e9
        step_bytes = sum(timer.bytes.values()) / args.steps
        # both roofs per conv family (their `work` is 2*MACs of the direct form): the larger fraction names the bound
        conv_fams = ('conv_igemm', 'conv_wino', 'conv_wino4')
        for k in conv_fams:
            if k in summ and timer.bytes.get(k):
                n_f, ms_f, fl_f = summ[k]
                mult = {'conv_igemm': 1.0, 'conv_wino': 2.25, 'conv_wino4': 4.0}[k]
                stages[k]['hbm_frac'] = round(timer.bytes[k] / (ms_f * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)
                stages[k]['mfma_frac'] = round(fl_f / mult / (ms_f * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4)
                stages[k]['bound'] = 'mfma' if stages[k]['mfma_frac'] > stages[k]['hbm_frac'] else 'hbm'
        hbm_frac = gbs / PEAK_HBM_GBS
        mfma_frac = stages[dom].get('mfma_frac', 0.0) if dom in conv_fams else 0.0
        by_mfma = mfma_frac > hbm_frac
        roofline = {'bound': 'mfma' if by_mfma else 'hbm',
                    'kernel': {'conv_igemm': ('conv_igemm_kernel + conv_igemm_b3_kernel (pointwise / dense convs: FP32 MFMA, and split-bf16 '
                                              'operands on BF16 MFMA; mfma_frac = algorithmic FLOPs / time vs the FP32-MFMA peak)') if merged
                               else 'conv_igemm_kernel (pointwise / dense convs, FP32 MFMA)',
                               'dwconv': 'dwconv kernels (depthwise k3/k5 + BN + swish)'}.get(dom, dom),
                    'achieved': round(mfma_frac * PEAK_FP32_MFMA_TFLOPS, 2) if by_mfma else round(gbs, 1),
                    'peak': PEAK_FP32_MFMA_TFLOPS if by_mfma else PEAK_HBM_GBS, 'unit': 'TFLOP/s' if by_mfma else 'GB/s',
                    'frac': round(max(hbm_frac, mfma_frac), 4), 'hbm_frac': round(hbm_frac, 4), 'mfma_frac': round(mfma_frac, 4),
                    'note': 'both roofs of the dominant family: hbm_frac = algorithmic bytes (operands once + result once) / HIP-event '
                            'time / 8 TB/s, mfma_frac = 2*MACs / time / 157.3 TFLOP/s; `bound` / `frac` = the larger one',
                    'step_algorithmic_bytes': round(step_bytes),
                    'step_achieved_GBs': round(step_bytes / (kernel_ms * 1e-3) / 1e9, 1),
                    'step_frac': round(step_bytes / (kernel_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
                    # what a maximally fused implementation would still move (block inputs / outputs, weights, residuals;
                    # not the tensors that live only inside an MBConv block or a separable conv): the stricter yardstick
                    'fused_min_bytes': round(sum(timer.fused.values()) / args.steps),
                    'fused_min_frac': round(sum(timer.fused.values()) / args.steps / (kernel_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)}
    # HBM/fabric bytes per launch from committed rocprofv3 --pmc passes of this same command (FETCH_SIZE / WRITE_SIZE in
    # separate passes, corrected per kernel as MI355X_MICROARCH.md prescribes); offline because counters need the profiler
    traffic, traffic_src = None, None
    tag = {'yolov3_80': 'yolov3', 'efficientdet-d1': 'd1', 'd1_fcs2_atss': 'fcos'}.get(args.config, args.config)
    for rnd in ('r06', 'r05', 'r04', 'r03', 'r02', 'r01'):
        name = f'{rnd}_pmc_traffic_{tag}_b{batch}_{args.size}.json' if rnd != 'r01' else 'r01_pmc_traffic_b32_640.json'
        path = os.path.join(ROOT, 'profiles', name)
        if os.path.exists(path) and (rnd != 'r01' or (args.config == 'yolov3_80' and batch == 32 and args.size == 640)):
            pmc_all = json.load(open(path))
            pmc = pmc_all.get(dom)
            if pmc and dom == 'conv_igemm' and args.config != 'yolov3_80' and pmc_all.get('conv_igemm_b3'):     # the merged family
                q = pmc_all['conv_igemm_b3']
                n_a, n_b = pmc.get('launches', 0), q.get('launches', 0)
                if n_a + n_b:
                    pmc = {'hbm_bytes_per_launch': (pmc.get('hbm_bytes_per_launch', 0) * n_a + q.get('hbm_bytes_per_launch', 0) * n_b) / (n_a + n_b)}
            if pmc:
                traffic = round(pmc.get('hbm_bytes_per_launch', pmc.get('hbm_read_bytes_per_launch_x2corr', 0) + pmc.get('hbm_write_bytes_per_launch', 0)))
                traffic_src = f'profiles/{name} (offline rocprofv3 --pmc passes' + (', an earlier round' if rnd != 'r06' else '') + ')'
                break
    roofline.update({'traffic': traffic, 'traffic_source': traffic_src,
                     'algorithmic_bytes_per_launch': round(timer.bytes.get(dom, 0.0) / n_k) if n_k else None,
                     'launches_per_step': n_k / args.steps, 'avg_launch_ms': round(ms_k / n_k, 4)})

    images = total * args.steps
    headline = args.config == 'yolov3_80'
    out = {
        'metric': (f'images/sec at batch {batch}, {args.size}x{args.size}, YOLOv3-80 (backbone -> FPN -> head -> decode -> NMS)'
                   if headline else f'images/sec at batch {batch}, {args.size}x{args.size}, {args.config} (backbone -> BiFPN -> head -> decode -> NMS)'),
        'value': round(images / elapsed, 2),
        'unit': 'images/sec',
        'n_gpus': world,
        'steps': args.steps,
        'warmup': args.warmup,
        'ms_per_step': round(elapsed / args.steps * 1e3, 3),
        'kernel_ms_per_step': round(kernel_ms, 3),
        'timer': ('HIP events on the launch stream; eager runs chain them (the event closing one launch opens the next), so '
                  'per-kernel times include inter-launch gaps and kernel_ms_per_step ~ ms_per_step by construction'
                  if not args.graph else 'HIP events on the launch stream around each launch of an eager re-run after the replayed timed '
                  'region (the same kernels: with batch lanes the lanes\' launch sequences one after the other on one stream)'),
        'higher_is_better': True,
        'scaling': 'weak',
        'vs_baseline': None,
        'dtype': 'f32',
        'arithmetic': ('float32 values and float32 accumulation throughout; ' +
                       ('direct convs with Cin % 16 == 0 form their float32 products from exact three-piece bfloat16 splits of both '
                        'operands (six v_mfma_f32_32x32x16_bf16 piece products per k-step; results equal the float32-instruction '
                        "kernel's to float32 round-off; MYDET_CONV_SPLIT_BF16=0 turns it off)" if ops.SPLIT_BF16 else
                        'every matrix product on v_mfma_f32_* (MYDET_CONV_SPLIT_BF16=0)')),
        'data': 'synthetic',
        'config': {'workload': f'{WORKLOADS.get(args.config, args.config)}, batch {batch}/GPU, {args.size}x{args.size}, '
                               'random-init calibrated weights' + (f', hipGraph replay ({graphed.lanes} batch lane{"s" if graphed.lanes > 1 else ""} per GPU)' if args.graph else ', eager launches'),
                   'global_batch': total, 'image_size': args.size, 'parallelism': f'dp{world}',
                   'batch_lanes': graphed.lanes if graphed is not None else 1,
                   'batch_lanes_rule': lanes_rule, 'batch_lanes_per_rank': lanes_per_rank if graphed is not None else [1] * world,
                   'passes': passes,
                   'exchange': f'one all-gather of {parallel.WORDS * 4} B detection records per image' if world > 1 else 'none'},
        'roofline': roofline,
        'stages': stages,
        'nms_p50_ms': stages['postprocess']['p50_ms'],
        'power': power,
    }
    n_lanes = graphed.lanes if graphed is not None else 1
    out['launches_per_lane'] = round(sum(v[0] for v in summ.values()) / args.steps / n_lanes, 1)
    if dist_on:
        out['rccl_ranks'] = torch.distributed.get_world_size()
        out['exchange_backend'] = str(torch.distributed.get_backend())
        # the line's n_gpus is a claim about what exchanged records: every rank of the RCCL group is one of the N GPUs
        assert out['rccl_ranks'] == world == args.gpus, (out['rccl_ranks'], world, args.gpus)
        assert out['exchange_backend'] == 'nccl', out['exchange_backend']
        out['rccl_ranks_equal_n_gpus'] = True
        out['exchange_ms_per_step'] = round(sum(a.elapsed_time(b) for a, b in exchange) / max(1, len(exchange)), 4)
        out['exchange_note'] = ('HIP events on the launch stream around the one all_gather_into_tensor of the step (rank 0; it includes '
                                'waiting for the slowest rank to arrive)')
    if verify is not None:
        out['verify'] = verify
    oracle_cand = None
    code = 0
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out['cpu_baseline'], oracle_cand = cpu_baseline(args.config, cfg, args.size, args.cpu_sample_batch, args.cpu_repeats, args.cpu_threads)
        out['cpu_baseline']['gpu_over_cpu'] = round(out['value'] / out['cpu_baseline']['value'], 1)
    elif world > 1:
        out['cpu_baseline'] = None
        out['cpu_baseline_note'] = 'the CPU baseline is timed at N = 1 only (rank 0 of a 1-GPU run)'
    if rank == 0:
        # parity gate of this very measurement: the timed step's records against the oracle.  The oracle's
        # candidates of the first images: the CPU baseline's sample, then one more oracle forward for the remaining ones
        import numpy as np
        have = 0 if oracle_cand is None else min(oracle_cand[0].shape[0], n_pc)
        more = oracle_candidates(args.config, cfg, args.size, lo + have, lo + n_pc, args.cpu_threads) if world == 1 else None
        if world == 1:
            parts = ([tuple(a[:have] for a in oracle_cand)] if have else []) + ([more] if more is not None else [])
            oracle_cand = tuple(np.concatenate([p_[j] for p_ in parts]) for j in range(3))
        extra_conf = sorted({0.05, float(cfg.get('test.default_conf_thres', 0.5))} - {float(conf)})
        out['parity_check'] = parity_check(pc_cand, pc_rec, conf, nms, oracle_cand, images=n_pc, extra_conf=extra_conf)
    if verify is not None and rank == 0 and not verify['ok']:
        code = 3
    if rank == 0 and not out['parity_check']['ok']:
        code = code or 4
    return (out if rank == 0 else None), code


if __name__ == '__main__':
    main()
