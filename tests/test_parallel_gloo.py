"""CPU, world_size 2 and 8 over gloo: the multi-GPU exchange (shard -> ONE all-gather of the fixed-size records -> views)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_records(rank, B):
    g = torch.Generator().manual_seed(100 + rank)
    count = torch.randint(0, 513, (B,), generator=g, dtype=torch.int32)
    return {'count': count, 'bbox': torch.rand(B, 512, 4, generator=g), 'class_idx': torch.randint(0, 80, (B, 512), generator=g),
            'score': torch.rand(B, 512, generator=g), 'index': torch.randint(0, 25200, (B, 512), generator=g, dtype=torch.int32)}


def _worker(rank, world, port, total, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from mydetection_amd import parallel
    lo, hi = parallel.shard_range(total, rank, world)
    glob = _fake_records(0, total)                       # the 1-process result for the whole batch
    rec = {k: v[lo:hi].clone() for k, v in glob.items()}  # this rank's shard
    allrec = parallel.gather_detections(rec, total=total)
    ok = allrec['count'].shape[0] == total
    for k in glob:
        ok = ok and torch.equal(allrec[k], glob[k])
    objs = parallel.records_to_objects(allrec, img_hw=(640, 640))
    ok = ok and len(objs) == total and all(len(o) == int(c) for o, c in zip(objs, allrec['count']))
    try:                                                  # a shard that is not this rank's range is rejected loudly
        parallel.gather_detections({k: v[:1] for k, v in rec.items()}, total=total + 2 * world)
        ok = False
    except ValueError:
        pass
    # the lane count bench.py --gpus N replays with: every rank's own application of the rule is compared -- the same
    # wish everywhere passes, ranks without an opinion (--lanes auto: only rank 0 timed anything) take rank 0's, and a
    # rank that wishes for something else makes EVERY rank raise (nobody is left waiting in the next collective)
    lanes, per_rank = parallel.agree_on_lanes(2)
    ok = ok and lanes == 2 and per_rank == [2] * world
    lanes, per_rank = parallel.agree_on_lanes(1 if rank == 0 else None)
    ok = ok and lanes == 1 and per_rank == [1] + [0] * (world - 1)
    try:
        parallel.agree_on_lanes(2 if rank != world - 1 else 1)
        ok = False
    except RuntimeError as e:
        ok = ok and 'disagree' in str(e)
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def _run_world(world, total):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
    assert res == [(r, True) for r in range(world)]


def test_gather_detections_world2():
    """gathered records == the single-process records of the same batch, even split."""
    _run_world(2, 6)


def test_gather_detections_world2_uneven():
    """7 images over 2 ranks (4 + 3): the short shard is padded for the collective and the padding dropped."""
    _run_world(2, 7)


def test_gather_detections_world8_batch256():
    """BASELINE configs[4]'s exchange shape on CPU: 256 images over 8 ranks (8 x 32), one all-gather of 16 400-B records."""
    _run_world(8, 256)


def test_gather_detections_world8_uneven_250():
    """250 images over 8 ranks (two shards of 32, six of 31): padded for the collective, padding rows dropped."""
    _run_world(8, 250)


def test_pack_unpack_roundtrip_and_sharding():
    from mydetection_amd import parallel
    rec = _fake_records(0, 4)
    out = parallel.make_records(rec)
    for k in rec:
        assert torch.equal(out[k], rec[k]), k
    again = parallel.record_views(out['records'].clone())
    for k in rec:
        assert torch.equal(again[k], rec[k]), k
    assert parallel.WORDS * 4 == 16400 and parallel.make_records(out) is out
    cover = []
    for r in range(8):
        lo, hi = parallel.shard_range(256, r, 8)
        assert hi - lo == 32
        cover += list(range(lo, hi))
    assert cover == list(range(256))
    sizes = [parallel.shard_range(10, r, 4) for r in range(4)]
    assert sizes == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert parallel.gather_detections(rec) is rec            # no process group: identity
    assert parallel.agree_on_lanes(2) == (2, [2])
