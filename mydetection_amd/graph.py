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
    """Captured `images -> detection records` for one (batch, H, W) and one (conf, nms) setting."""

    def __init__(self, model, example, conf_thres, nms_thres, warmup=2):
        assert example.is_cuda and example.dim() == 4
        self.model = model
        self.static_in = example.clone()
        self.conf, self.nms = float(conf_thres), float(nms_thres)
        self.epoch = getattr(model, 'weights_epoch', 0)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(warmup):                       # first launches set function attributes, fill caches
                self._run_eager()
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph), torch.no_grad():
            self.cand, self.records = self._run_eager()
        # the warm-up sized every scratch buffer and prepared every parameter (a growth during capture raises), so what
        # exists now is exactly what the graph recorded
        self._held = (ops.live_workspaces(example.device), prepared_tensors(model))

    def _run_eager(self):
        bb, ci, sc = self.model.forward_candidates(self.static_in)
        return (bb, ci, sc), batched_post_process(bb, ci, sc, self.conf, self.nms)

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
