"""Reproducer hunt for the round-5 squeeze-excite tail finding, outside the model: the depthwise launch with the in-launch gate
(1152 channels, 5x5, 20x20, batch 16) loops on stream A while stream B loops an aggressor; every gate is compared with the gate of
the same launch run alone.  Build with -DMYDET_SE_PK for the compiler's v_pk_fma_f32 form of the expand conv (diagnostic)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mydetection_amd import ops
from mydetection_amd.external.efficientnet.model import static_same_pad
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(5)
B, C, Cse, k, H = 16, 1152, 48, 5, 20
x = torch.randn(B, C, H, H, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
wd = (torch.randn(k, k, C, generator=g) / k).to(dev)
sc, sh = (torch.rand(C, generator=g) + 0.5).to(dev), (torch.randn(C, generator=g) * 0.3).to(dev)
w1 = (torch.randn(Cse, C, generator=g) / C ** 0.5).to(dev)
b1 = (torch.randn(Cse, generator=g) * 0.1).to(dev)
w2t = (torch.randn(Cse, C, generator=g) * 0.3).to(dev)
b2 = (torch.randn(C, generator=g) * 0.1).to(dev)
pad = static_same_pad(k, 1, 640 // 32 * 1)
se = (w1, b1, w2t, b2)
_, ref = ops.dwconv(x, wd, sc, sh, k, 1, (2, 2, 2, 2), ops.ACT_SWISH, se=se)
ref = ref.clone()
torch.cuda.synchronize()
# aggressors on stream B
xa = torch.randn(B, 192, H, H, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
wa = (torch.randn(1152, 1, 1, 192, generator=g) / 192 ** 0.5).to(dev)
sa, ha = (torch.rand(1152, generator=g) + 0.5).to(dev), (torch.randn(1152, generator=g) * 0.1).to(dev)
w3 = ops.split_bf16(wa)
x2 = torch.randn(32, 128, 160, 160, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
w2 = (torch.randn(256, 3, 3, 128, generator=g) / (128 * 9) ** 0.5).to(dev)
s2, h2 = (torch.rand(256, generator=g) + 0.5).to(dev), (torch.randn(256, generator=g) * 0.1).to(dev)
w23 = ops.split_bf16(w2)
aggr = {'none': None,
        'split-bf16 192->1152 1x1 @20^2 (the layer of the finding)': lambda: ops.conv2d(xa, wa, sa, ha, 1, 1, (0, 0, 0, 0), ops.ACT_SWISH, b3=w3, b3_min_rows=1),
        'float32 192->1152 1x1 @20^2': lambda: ops.conv2d(xa, wa, sa, ha, 1, 1, (0, 0, 0, 0), ops.ACT_SWISH),
        'split-bf16 128->256 s2 @160^2 (long launches)': lambda: ops.conv2d(x2, w2, s2, h2, 3, 2, (1, 1, 1, 1), ops.ACT_LEAKY, b3=w23, b3_min_rows=1)}
import ctypes
from mydetection_amd import _lib
sink = torch.zeros(16, device=dev)


def busy(kind, iters, lds, blocks=1024):
    def f():
        h = _lib.lib()
        if hasattr(h, 'mydet_diag_busy'):
            h.mydet_diag_busy(ctypes.c_int(kind), ctypes.c_int(iters), ctypes.c_int(lds), ctypes.c_int(blocks), ctypes.c_void_p(sink.data_ptr()),
                              ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    return f


if hasattr(_lib.lib(), 'mydet_diag_busy'):
    aggr.update({'synthetic: bf16 MFMA loop, 48 KB LDS per workgroup': busy(0, 4000, 48 * 1024),
                 'synthetic: bf16 MFMA loop, no LDS request': busy(0, 4000, 0),
                 'synthetic: bf16 MFMA fed by ds_read_b128, 48 KB': busy(2, 4000, 48 * 1024),
                 'synthetic: float32 MFMA loop, 48 KB': busy(1, 1000, 48 * 1024),
                 'synthetic: float32 -> bf16 piece arithmetic (VALU), 48 KB': busy(3, 4000, 48 * 1024),
                 'synthetic: SIX independent bf16 32x32x16 accumulators, 48 KB': busy(4, 1500, 48 * 1024),
                 'synthetic: SIX independent bf16 32x32x16 accumulators, no LDS': busy(4, 1500, 0),
                 'synthetic: six independent float32 32x32x2 accumulators, 48 KB': busy(5, 400, 48 * 1024),
                 'synthetic: six independent bf16 16x16x32 accumulators, 48 KB': busy(6, 3000, 48 * 1024)})
def b3_dbg(bits):
    base = aggr['split-bf16 128->256 s2 @160^2 (long launches)']

    def f():
        os.environ['MYDET_B3_DBG'] = str(bits)
        base()
        os.environ['MYDET_B3_DBG'] = '0'
    return f


for bits, what in ((4, 'no MFMA'), (1 + 2 + 16, 'MFMA on stale LDS only')):
    aggr[f'split-bf16 s2 @160^2 minus: {what}'] = b3_dbg(bits)
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
hdr = ops.se_shares(dev, 1).view(torch.int32)
for epoch0 in (None, 0x3F800000):
  if epoch0 is not None:
    hdr[0] = epoch0
    print(f'--- launch counter set to 0x{epoch0:08X}: the epoch tags left in registers by the poll loop are NORMAL floats now (before: small integers = float32 denormals)', flush=True)
  for name, fn in aggr.items():
      if epoch0 is not None and not (name.startswith('split-bf16 128') or '16x16x32' in name or name == 'none'):
          continue
      with ops.lane(1):                                   # the aggressor's scratch is another lane's
          if fn is not None:
              with torch.cuda.stream(sB):
                  fn()
      torch.cuda.synchronize()
      bad = runs = 0
      worst = 0.0
      t_end = time.time() + 2.5
      while time.time() < t_end:
          gates = []
          if fn is not None:
              with torch.cuda.stream(sB), ops.lane(1):
                  for _ in range(60 if 'synthetic' not in name else 12):
                      fn()
          with torch.cuda.stream(sA):
              for _ in range(40):
                  gates.append(ops.dwconv(x, wd, sc, sh, k, 1, (2, 2, 2, 2), ops.ACT_SWISH, se=se)[1])
          torch.cuda.synchronize()
          for gt in gates:
              runs += 1
              d = (gt - ref).abs().max().item()
              if d != 0:
                  bad += 1
                  worst = max(worst, d)
      print(f'aggressor {name:60s}: {bad} of {runs} gates differ from the launch run alone (worst {worst:.3e})', flush=True)
