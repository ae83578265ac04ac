"""CPU, world_size 2 over gloo: the multi-GPU exchange (shard -> pack records -> ONE all-gather -> unpack)."""
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


def _worker(rank, world, port, B, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from mydetection_amd import parallel
    rec = _fake_records(rank, B)
    allrec = parallel.gather_detections(rec)
    ok = allrec['count'].shape[0] == world * B
    for r in range(world):
        ref = _fake_records(r, B)
        for k in ref:
            ok = ok and torch.equal(allrec[k][r * B:(r + 1) * B], ref[k])
    objs = parallel.records_to_objects(allrec, img_hw=(640, 640))
    ok = ok and len(objs) == world * B and all(len(o) == int(c) for o, c in zip(objs, allrec['count']))
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_gather_detections_world2():
    world, B = 2, 3
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, B, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
    assert res == [(0, True), (1, True)]


def test_pack_unpack_roundtrip_and_sharding():
    from mydetection_amd import parallel
    rec = _fake_records(0, 4)
    out = parallel.unpack_records(parallel.pack_records(rec))
    for k in rec:
        assert torch.equal(out[k], rec[k]), k
    assert parallel.WORDS * 4 == 14340
    cover = []
    for r in range(8):
        lo, hi = parallel.shard_range(256, r, 8)
        assert hi - lo == 32
        cover += list(range(lo, hi))
    assert cover == list(range(256))
    sizes = [parallel.shard_range(10, r, 4) for r in range(4)]
    assert sizes == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert parallel.gather_detections(rec) is rec            # no process group: identity
