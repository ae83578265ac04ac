"""Data-parallel inference over the GPUs of one node (no reference counterpart: the reference is
single-process, single-device).

Images are independent (`post_process` is per image, utils/structures.py:92), so a batch is split
into contiguous shards, one process per GPU runs backbone -> NMS locally, and the only exchange is
ONE all-gather (RCCL over xGMI; `nccl` backend) of fixed-size detection records.  The post-process
kernel writes those records itself (include/mydet.h, MYDET_REC_*):
    per image  count:i32 +3 pad | 512 x (cx,cy,w,h:f32) | 512 x score:f32 | 512 x class:i64 | 512 x index:i32
  = 4100 words = 16 400 B.  32 images/GPU -> 525 KB per rank: latency-bound, one collective, no reduce, and
no pack/unpack pass on either side -- the dict the package works with is a set of views of that buffer.
"""
import torch
import torch.distributed as dist

from . import _lib

TOPK = _lib.REC_TOPK
WORDS = _lib.REC_WORDS


def shard_range(total, rank, world):
    """Contiguous shard [lo, hi) of `total` images for `rank`; sizes differ by at most one."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def record_views(records):
    """Field views of an int32 [B, WORDS] record buffer (any device)."""
    from .ops import record_views as views
    return views(records)


def make_records(rec):
    """dict of separate tensors (count, bbox, score, class_idx, index) -> record dict backed by one buffer.
    Only for records that did not come from `ops.postprocess` (tests, host-side tools)."""
    if 'records' in rec:
        return rec
    B = rec['count'].shape[0]
    buf = torch.zeros((B, WORDS), dtype=torch.int32, device=rec['count'].device)
    out = record_views(buf)
    for k in ('count', 'bbox', 'score', 'class_idx', 'index'):
        out[k].copy_(rec[k])
    return out


def gather_detections(rec, group=None, always=False, total=None):
    """All ranks end up with the records of the whole batch in image order.  `rec`: this rank's shard (contiguous,
    `shard_range`).  Shards of unequal size are padded to the largest one for the collective and the padding rows
    dropped afterwards; `total` (the global image count) is required then and checked otherwise.
    `always` runs the collective even for a one-rank group (rehearses the RCCL path on one GPU)."""
    if not (dist.is_available() and dist.is_initialized()):
        return rec
    world = dist.get_world_size(group)
    if world == 1 and not always:
        return rec
    rank = dist.get_rank(group)
    buf = make_records(rec)['records']
    B = buf.shape[0]
    if total is None:
        total = B * world                   # equal shards; a mismatch across ranks fails in the collective's size check
    lo, hi = shard_range(total, rank, world)
    if hi - lo != B:
        raise ValueError(f'rank {rank} holds {B} images but shard_range({total}, {rank}, {world}) is [{lo}, {hi})')
    cap = -(-total // world)                # largest shard
    if B < cap:
        buf = torch.cat([buf, buf.new_zeros((cap - B, WORDS))])
    out = torch.empty((world * cap, WORDS), dtype=torch.int32, device=buf.device)
    dist.all_gather_into_tensor(out, buf, group=group)
    if total != world * cap:                # drop the padding row of the short shards
        keep = torch.cat([torch.arange(r * cap, r * cap + (lambda a: a[1] - a[0])(shard_range(total, r, world)))
                          for r in range(world)]).to(out.device)
        out = out.index_select(0, keep)
    return record_views(out)


def agree_on_lanes(choice, device=None, group=None):
    """One batch-lane count for every rank.  Every rank passes what it would replay with (its own application of the
    shared rule) or None when it has no opinion (`--lanes auto`: only rank 0 timed anything); the wishes are
    all-gathered, rank 0's counts, and a rank whose own wish differs raises -- on every rank, so nobody is left
    waiting in a later collective.  Lanes change the last float bits of a result (graph.GraphedPath), so ranks that
    replayed different counts would disagree bit for bit with a 1-GPU recomputation (`bench.py --verify`).
    Returns (lanes, [wish of each rank, 0 = none]).  No process group: (choice, [choice])."""
    if not (dist.is_available() and dist.is_initialized()):
        return int(choice), [int(choice)]
    world = dist.get_world_size(group)
    mine = torch.tensor([0 if choice is None else int(choice)], dtype=torch.int32, device=device)
    got = [torch.zeros(1, dtype=torch.int32, device=device) for _ in range(world)]
    dist.all_gather(got, mine, group=group)
    wishes = [int(t.item()) for t in got]
    if wishes[0] < 1:
        raise RuntimeError(f'rank 0 gave no batch-lane count: {wishes}')
    if any(w not in (0, wishes[0]) for w in wishes):
        raise RuntimeError(f'ranks disagree on the batch-lane count: {wishes} (MYDET_LANES / --lanes must be the same on every rank)')
    return wishes[0], wishes


def records_to_objects(rec, img_hw=None, bb_format='cxcywh'):
    """Fixed-size records -> List[ImageObjects] (one host sync for the counts)."""
    from .utils.structures import ImageObjects
    from .ops import check_counts
    counts = check_counts(rec['count'].cpu().tolist())
    return [ImageObjects(rec['bbox'][b, :k], rec['class_idx'][b, :k], None, rec['score'][b, :k], bb_format, img_hw)
            for b, k in enumerate(counts)]
