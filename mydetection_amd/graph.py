"""hipGraph capture of the whole hot path for a fixed input shape.

A forward is 83 (YOLOv3) to ~290 (EfficientDet-D1) kernel launches issued from Python; at small
batch the host cannot issue them as fast as the GPU retires them.  Every C-ABI entry point only
enqueues work on the stream it is given (include/mydet.h), so the sequence
    model.forward_candidates(x) -> batched_post_process(...)
is captured once into a HIP graph (torch.cuda.CUDAGraph is the hipGraph wrapper) and replayed with
one host call per batch.  Buffers are owned by the graph's private pool; the input is a static
tensor that `__call__` copies into (device-to-device when the caller's batch is already in HBM).
"""
import torch

from .utils.structures import batched_post_process


class GraphedPath:
    """Captured `images -> detection records` for one (batch, H, W) and one (conf, nms) setting."""

    def __init__(self, model, example, conf_thres, nms_thres, warmup=2):
        assert example.is_cuda and example.dim() == 4
        self.model = model
        self.static_in = example.clone()
        self.conf, self.nms = float(conf_thres), float(nms_thres)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(warmup):                       # first launches set function attributes, fill caches
                self._run_eager()
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph), torch.no_grad():
            self.cand, self.records = self._run_eager()

    def _run_eager(self):
        bb, ci, sc = self.model.forward_candidates(self.static_in)
        return (bb, ci, sc), batched_post_process(bb, ci, sc, self.conf, self.nms)

    def __call__(self, x=None):
        """Replay.  Returns the static record dict (count/bbox/class_idx/score/index), overwritten by the
        next call; `x` (same shape) is copied into the static input first when given."""
        if x is not None and x.data_ptr() != self.static_in.data_ptr():
            self.static_in.copy_(x, non_blocking=True)
        self.graph.replay()
        return self.records
