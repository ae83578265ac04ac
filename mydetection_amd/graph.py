"""hipGraph capture of the whole hot path for a fixed input shape.

A forward is 83 (YOLOv3) to ~290 (EfficientDet-D1) kernel launches issued from Python; at small
batch the host cannot issue them as fast as the GPU retires them.  Every C-ABI entry point only
enqueues work on the stream it is given (include/mydet.h), so the sequence
    model.forward_candidates(x) -> batched_post_process(...)
is captured once into a HIP graph (torch.cuda.CUDAGraph is the hipGraph wrapper) and replayed with
one host call per batch.  Buffers are owned by the graph's private pool; the input is a static
tensor that `__call__` copies into (device-to-device when the caller's batch is already in HBM).

Lifetime rule: a captured graph replays raw device addresses.  Activations live in the graph's own
pool, but two kinds of memory it touches are owned elsewhere and can be replaced behind its back:
the per-device scratch buffers of `ops` (the F(4x4) workspace is swapped for a larger one when a
bigger layer shows up) and the kernel-ready parameter copies cached on the modules (rebuilt after
`load_state_dict`).  `GraphedPath` therefore keeps a reference to every such tensor that existed
when it was captured (`_held`): a superseded buffer stays allocated until the last graph that
recorded its address is dropped, so a replay can never write into memory the caching allocator has
handed to another tensor.  A graph captured before a parameter change would still compute with the
OLD parameters; `stale()` tells (the model's `weights_epoch` moves on `load_state_dict` / `.to()`),
and `GraphCache` / `api.Detector` drop such graphs instead of replaying them.
"""
import os
from collections import OrderedDict

import torch

from . import ops
from .utils.structures import batched_post_process


def prepared_tensors(model):
    """Every kernel-ready parameter copy cached on the model's modules at this moment (models.modules.prepare_conv
    and friends keep them in `_prep_cache` / `_w2t` dicts): the tensors a captured launch sequence reads."""
    held = []
    for m in model.modules():
        for slot in ('_prep_cache', '_w2t'):
            cache = m.__dict__.get(slot)
            if cache:
                held.append(dict(cache))        # shallow copy: keeps the current values alive when a slot is rebuilt
    return held


class GraphedPath:
    """Captured `images -> detection records` for one (batch, H, W) and one (conf, nms) setting.

    `lanes`: the images of a batch are independent, so the batch can be cut into equal parts that run the whole path on
    streams of their own (fork / join inside the capture -> parallel branches of the hipGraph).  The hardware
    dispatcher then fills the CUs that one lane's launch leaves idle -- its ramp, its tail, the small grids of the
    coarse pyramid levels -- with the other lane's workgroups.  Measured (profiles/r03_lanes.md): two lanes are +5 % on
    the EfficientDet-family configs (~140 launches of 10-150 us per step) and -5 % on YOLOv3 (launches of 0.3-1 ms that
    fill the chip on their own), so the default `'auto'` (env MYDET_LANES) captures the path with one and with two
    lanes, times a few replays of each and keeps the faster graph.  Each lane has its own scratch (`ops.lane`).
    A lane's launches see a batch of B / lanes images, so float results can differ from a full-batch eager pass in the
    last bits (grid-size dependent K cuts, exactly as a solo image differs from the same image in a batch);
    `eager()` runs the same decomposition without the graph and is bit-identical to a replay."""

    def __init__(self, model, example, conf_thres, nms_thres, warmup=2, lanes=None):
        assert example.is_cuda and example.dim() == 4
        self.model = model
        self.static_in = example.clone()
        self.conf, self.nms = float(conf_thres), float(nms_thres)
        self.epoch = getattr(model, 'weights_epoch', 0)
        if lanes is None:
            lanes = os.environ.get('MYDET_LANES', 'auto')
        B = example.shape[0]
        if lanes == 'auto':
            tries = [1, 2] if B >= 2 and B % 2 == 0 else [1]
        else:
            tries = [int(lanes) if 1 < int(lanes) <= B else 1]         # a forced count may cut unevenly (tensor_split)
        best = None
        for n in tries:
            cap = self._capture(n, warmup)
            if len(tries) > 1:
                cap['ms'] = self._time_replays(cap['graph'])
                if best is not None and cap['ms'] >= best['ms']:
                    del cap
                    continue
            best = cap
        self.lanes, self._streams = best['lanes'], best['streams']
        self.graph, self._cand, self.records = best['graph'], best['cand'], best['records']
        self.tuned_ms = best.get('ms')
        # the warm-up sized every scratch buffer and prepared every parameter (a growth during capture raises), so what
        # exists now is exactly what the graph recorded
        self._held = (ops.live_workspaces(example.device), prepared_tensors(model))

    def _capture(self, lanes, warmup):
        self.lanes = lanes
        self._streams = [torch.cuda.Stream() for _ in range(lanes)] if lanes > 1 else []
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(warmup):                       # first launches set function attributes, fill caches
                self._run_eager()
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph), torch.no_grad():
            cand, records = self._run_eager()
        return dict(lanes=lanes, streams=self._streams, graph=graph, cand=cand, records=records)

    @staticmethod
    def _time_replays(graph, n=5):
        graph.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            graph.replay()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / n

    def _run_eager(self):
        if self.lanes == 1:
            bb, ci, sc = self.model.forward_candidates(self.static_in)
            return (bb, ci, sc), batched_post_process(bb, ci, sc, self.conf, self.nms)
        main = torch.cuda.current_stream()
        parts = self.static_in.tensor_split(self.lanes)
        # every lane writes its rows of ONE record buffer (allocated on the joining stream); the candidates stay per lane
        records = torch.empty((self.static_in.shape[0], ops._lib.REC_WORDS), dtype=torch.int32, device=self.static_in.device)
        cands, lo = [], 0
        for i, (st, part) in enumerate(zip(self._streams, parts)):
            st.wait_stream(main)
            with torch.cuda.stream(st), ops.lane(i):
                bb, ci, sc = self.model.forward_candidates(part)
                batched_post_process(bb, ci, sc, self.conf, self.nms, records=records[lo:lo + part.shape[0]])
                cands.append((bb, ci, sc))
            lo += part.shape[0]
        for st in self._streams:
            main.wait_stream(st)
        return cands, ops.record_views(records)

    def eager(self, x=None):
        """The captured launch sequence issued from the host (same lanes, same streams): what a replay computes, bit for
        bit, in fresh tensors."""
        if x is not None and x.data_ptr() != self.static_in.data_ptr():
            self.static_in.copy_(x, non_blocking=True)
        with torch.no_grad():
            return self._run_eager()[1]

    @property
    def cand(self):
        """(bbox, class_idx, score) of the last replay, before post-processing (joined on demand when lanes > 1)."""
        if self.lanes == 1:
            return self._cand
        return tuple(torch.cat([c[j] for c in self._cand]) for j in range(3))

    def stale(self):
        """True when the model's parameters were replaced after the capture (the graph would compute with the old ones)."""
        return getattr(self.model, 'weights_epoch', 0) != self.epoch

    def __call__(self, x=None):
        """Replay.  Returns the static record dict (count/bbox/class_idx/score/index), overwritten by the
        next call; `x` (same shape) is copied into the static input first when given."""
        if self.stale():
            raise RuntimeError('GraphedPath: the model parameters changed after this graph was captured; capture a new one')
        if x is not None and x.data_ptr() != self.static_in.data_ptr():
            self.static_in.copy_(x, non_blocking=True)
        self.graph.replay()
        return self.records


class GraphCache:
    """LRU of captured graphs keyed by (input shape, thresholds), with the capture policy of `api.Detector`:

    * a key is captured once it has been served eagerly `need` times (1 to start with: a one-off shape is not worth
      two warm-up passes and a private activation pool);
    * a hit moves the entry to the young end; a capture into a full cache evicts the oldest entry, forgets how often
      that key was seen and DOUBLES `need` (up to 1024): a workload that cycles through more shapes than the cache
      holds -- 'pad_divisible' preprocessing over a mixed-size image set -- soon stops capturing and runs eagerly
      instead of paying warm-up + capture + replay on most calls;
    * entries captured before the model's parameters changed are dropped on lookup.
    Pure bookkeeping (no GPU call): `make` builds the graph for a key."""

    def __init__(self, capacity=6):
        self.capacity = max(1, int(capacity))
        self.graphs = OrderedDict()
        self.seen = {}
        self.need = 1
        self.captures = self.evictions = 0

    def lookup(self, key):
        g = self.graphs.get(key)
        if g is not None:
            if getattr(g, 'stale', lambda: False)():
                del self.graphs[key]
                return None
            self.graphs.move_to_end(key)
        return g

    def should_capture(self, key):
        """Call when `key` is about to be served eagerly; True when it has earned a graph."""
        return self.seen.get(key, 0) >= self.need

    def note_eager(self, key):
        self.seen[key] = self.seen.get(key, 0) + 1
        if len(self.seen) > 4096:                       # unbounded variety of shapes: forget the counts, keep the graphs
            self.seen = {k: v for k, v in self.seen.items() if k in self.graphs}

    def insert(self, key, graph):
        while len(self.graphs) >= self.capacity:
            old, _ = self.graphs.popitem(last=False)
            self.seen.pop(old, None)
            self.evictions += 1
            self.need = min(self.need * 2, 1024)
        self.graphs[key] = graph
        self.captures += 1
        return graph

    def clear(self):
        self.graphs.clear()
        self.seen.clear()
