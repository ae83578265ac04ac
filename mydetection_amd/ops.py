"""Torch-tensor front end of the C ABI (include/mydet.h).

PyTorch is plumbing here: tensors are HBM allocations plus the current HIP stream.
Activations are logical NCHW tensors stored channels-last (physical [B,H,W,ld]) so
module inputs/outputs have the reference's shapes while kernels see NHWC.
Every function launches hand-written HIP kernels; no arithmetic falls back to ATen (a
input in a foreign layout is re-laid out with one tensor copy before the launch).
"""
import ctypes
import os
import threading

import numpy as np
import torch

from . import _lib

ACT_NONE, ACT_LEAKY, ACT_SWISH = 0, 1, 2
DECODE_YOLO, DECODE_RETINA, DECODE_FCOS = 0, 1, 2
TOPK = 512


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


class KernelTimer:
    """Optional per-launch timing with HIP events recorded on the launch stream (bench.py uses it to
    price the dominant kernel inside the timed region).  Disabled (None) by default: zero overhead."""

    def __init__(self, chain=False):
        self.spans = {}
        self.bytes = {}                  # name -> algorithmic bytes (operands read once + result written once)
        # name -> bytes a maximally fused implementation would still move: tensors that exist only between two layers of a
        # block the reference itself treats as a unit (the 6x-wide maps inside an MBConv block, the depthwise result
        # inside a separable conv) are not counted; block inputs / outputs, weights and residuals are
        self.fused = {}
        # chain=True: the event that closes one launch also opens the next one (every launch of the step is timed and
        # they run back to back on one in-order stream, so "end of launch i" IS "start of launch i+1"): half the
        # event records -- each is a small packet on the GPU's queue -- inside the timed region
        self.chain = chain
        self._tail = None

    def start(self):
        if self.chain and self._tail is not None:
            return self._tail
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()                      # torch's current stream == the stream handed to the C ABI
        return ev

    def stop(self, name, start_ev, work=0.0, nbytes=0.0, fused=None):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        self._tail = ev
        self.spans.setdefault(name, []).append((start_ev, ev, work))
        self.bytes[name] = self.bytes.get(name, 0.0) + nbytes
        self.fused[name] = self.fused.get(name, 0.0) + (nbytes if fused is None else fused)

    def cut(self):
        """Forget the chain (call where untimed GPU work, a synchronisation or a step boundary intervenes)."""
        self._tail = None

    def summary(self):
        """name -> (launches, total_ms, total_work); call after torch.cuda.synchronize()."""
        return {k: (len(v), sum(a.elapsed_time(b) for a, b, _ in v), sum(w for _, _, w in v))
                for k, v in self.spans.items()}


TIMER = None        # set to a KernelTimer() to record
TIMER_DETAIL = False  # name conv launches by shape (tools/profile_layers.py)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def require_gpu(t, what):
    if not t.is_cuda:
        raise RuntimeError(f'{what}: tensor is on {t.device}; mydetection_amd runs on MI355X only '
                           '(there is no CPU path in this package)')


def nhwc_ld(x):
    """Pixel stride of a logical [B,C,H,W] tensor stored as [B,H,W,ld], or None if it is not."""
    B, C, H, W = x.shape
    sb, sc, sh, sw = x.stride()
    if x.dtype != torch.float32 or (C > 1 and sc != 1):
        return None
    ld = sw if W > 1 else (sh if H > 1 else (sb if B > 1 else C))
    if ld < C or ld % 4 or (W > 1 and sw != ld) or (H > 1 and sh != W * ld) or (B > 1 and sb != H * W * ld):
        return None
    if x.data_ptr() % 16:
        return None
    return ld


def to_nhwc(x):
    """Return (tensor, ld) with the tensor in a kernel-readable channels-last layout."""
    ld = nhwc_ld(x)
    if ld is not None:
        return x, ld
    B, C, H, W = x.shape
    Cp = (C + 3) // 4 * 4
    buf = x.new_zeros((B, H, W, Cp)) if Cp != C else x.new_empty((B, H, W, Cp))
    buf[..., :C] = x.permute(0, 2, 3, 1)
    return buf.permute(0, 3, 1, 2)[:, :C], Cp


def empty_nhwc(B, C, H, W, device, ld=None):
    ld = ld or (C + 3) // 4 * 4
    buf = torch.empty((B, H, W, ld), dtype=torch.float32, device=device)
    return buf.permute(0, 3, 1, 2)[:, :C], ld


_WORKSPACE = {}
WORKSPACE_BYTES = 64 << 20
_LANE = 0


class lane:
    """`with ops.lane(i):` -- launches issued inside use lane i's scratch buffers.  Scratch is reused in stream order,
    so two launch sequences that run on different streams at the same time (graph.GraphedPath's batch lanes) must not
    share it; everything outside a `with` is lane 0."""
    def __init__(self, index):
        self.index = int(index)

    def __enter__(self):
        global _LANE
        self.prev, _LANE = _LANE, self.index

    def __exit__(self, *exc):
        global _LANE
        _LANE = self.prev


def _scratch_key(device):
    """(device, lane, host thread): scratch buffers are reused in stream order by ONE launch sequence, so a second host thread that
    happens to be on the same lane index gets buffers of its own.  Not keyed by stream: a hipGraph is warmed
    up on one stream and captured on another, and must find the buffers its warm-up sized (growth during capture raises)."""
    return (device.type, device.index, _LANE, threading.get_ident())


def conv_workspace(device):
    """Per-device, per-lane, per-stream scratch for the split-K tail of conv2d (stream-ordered reuse)."""
    key = _scratch_key(device)
    if key not in _WORKSPACE:
        _WORKSPACE[key] = torch.empty(WORKSPACE_BYTES // 4, dtype=torch.float32, device=device)
    return _WORKSPACE[key]


_WINO4_WS = {}


def wino4_workspace(device, nbytes):
    """Per-device, per-lane scratch for the transform-domain input of the F(4x4,3x3) kernel: grows to the largest layer
    seen (stream-ordered reuse; one stream per lane).  A hipGraph must be captured after an eager pass has sized it.
    Growth REPLACES the buffer: whoever recorded its address (a captured hipGraph) keeps the superseded tensor alive
    through `live_workspaces` -- graph.GraphedPath does -- so a replay never writes into memory the caching allocator
    has handed to someone else."""
    key = _scratch_key(device)
    ws = _WINO4_WS.get(key)
    if ws is None or ws.numel() * 4 < nbytes:
        if device.type == 'cuda' and torch.cuda.is_current_stream_capturing():
            raise RuntimeError('wino4_workspace: the workspace would grow during stream capture; run the model once eagerly first')
        ws = _WINO4_WS[key] = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=device)
    return ws


_SE_SHARES = {}
SE_MAX_CSE = 96
# MYDET_SE_IN_DW=0: the squeeze-excite gate as a launch of its own (mydet_se_gate_f32) behind every depthwise launch (A/B)
SE_IN_DW = os.environ.get('MYDET_SE_IN_DW', '1') != '0'
# The in-launch tail is used from this many (expanded) channels up.  Measured per block on a lane of 8 images
# (profiles/r05_se_in_dw.txt: depthwise + gate launch vs depthwise launch with the tail): 1152 channels 34.5 -> 25.3 us,
# 1920 channels 48.0 -> 36.6 us -- the gate launch is a 13-32 us chain of dependent round trips on one CU per image there --,
# but +-1 us at 16 .. 672 channels (the gate launch costs 7-10 us, the finishing workgroup about as much) and 3 us WORSE in
# the fused expand + depthwise launches, so the narrow blocks keep the separate launch.  MYDET_SE_IN_DW_MIN_C overrides.
SE_IN_DW_MIN_C = int(os.environ.get('MYDET_SE_IN_DW_MIN_C', '1000'))


def se_shares(device, n_pairs):
    """Per-device, per-lane share buffer of the in-launch squeeze-excite tail (include/mydet.h: mydet_se_tail.hpart): a header
    (launch counter = 1, finished-image count = 0) followed by room for `n_pairs` (value, epoch) pairs (0).  The kernels' own protocol keeps it consistent, so
    ONE buffer serves every layer of a lane's stream.  Grows to the largest layer seen (not during a stream capture)."""
    key = _scratch_key(device)
    buf = _SE_SHARES.get(key)
    need = _lib.SE_EPOCH_WORDS + 2 * int(n_pairs)
    if buf is None or buf.numel() < need:
        if device.type == 'cuda' and torch.cuda.is_current_stream_capturing():
            raise RuntimeError('se_shares: the share buffer would grow during stream capture; run the model once eagerly first')
        buf = torch.zeros(max(need, 1 << 21), dtype=torch.int32, device=device)
        buf[0] = 1                                           # the launch counter; word 1 = images finished in a running launch
        _SE_SHARES[key] = buf = buf.view(torch.float32)
    return buf


def se_tail_timeouts(device):
    """Number of in-launch squeeze-excite tails (summed over this thread's lanes' share buffers) whose finishing workgroup gave up
    waiting for a share and wrote a NaN gate (header word 2 of the share buffer, csrc/se_tail.h): 0 unless a launch was starved
    for ~2^14 polls.  Synchronises."""
    n = 0
    for k, buf in _SE_SHARES.items():
        if k[:2] == (device.type, device.index):
            n += int(buf.view(torch.int32)[2].item())
    return n


def _se_tail(se, B, C, groups, device):
    """(ctypes struct, gate tensor, tensors to keep alive) for `se` = (w1 [Cse,C], b1, w2t [Cse,C], b2) or (None, None, ())."""
    if se is None:
        return None, None, ()
    w1, b1, w2t, b2 = se
    Cse = w1.shape[0]
    assert tuple(w1.shape) == (Cse, C) and tuple(w2t.shape) == (Cse, C) and w1.is_contiguous() and w2t.is_contiguous()
    assert Cse <= SE_MAX_CSE and groups > 0
    gate = torch.empty((B, C), dtype=torch.float32, device=device)
    hpart = se_shares(device, B * groups * Cse)
    t = _lib.SeTail(w1.data_ptr(), b1.data_ptr(), w2t.data_ptr(), b2.data_ptr(), gate.data_ptr(), hpart.data_ptr(), Cse, hpart.numel() * 4)
    return t, gate, (hpart, w1, b1, w2t, b2)


def live_workspaces(device):
    """The scratch tensors launches on `device` are currently handed (split-K workspace, F(4x4) transform-domain
    input).  A captured launch sequence holds on to this list for as long as it can be replayed."""
    key = (device.type, device.index)
    return [t for d in (_WORKSPACE, _WINO4_WS, _SE_SHARES) for k, t in d.items() if k[:2] == key]


def conv_out_size(n, k, stride, pad_lo, pad_hi):
    return (n + pad_lo + pad_hi - k) // stride + 1


# MYDET_CONV_WINO=0 keeps every 3x3 layer on the direct implicit-GEMM kernel (A/B measurements)
WINOGRAD = os.environ.get('MYDET_CONV_WINO', '1') != '0'


def wino_weights(w_ohwi):
    """Transform-domain copy of a 3x3 OHWI weight for `conv2d(..., wino=)`, or None when the shape is not covered
    (Cin % 8, Cout % 4) -- the layer then stays on the direct kernel."""
    require_gpu(w_ohwi, 'wino_weights')
    Cout, kh, kw, Cin = w_ohwi.shape
    if kh != 3 or kw != 3 or Cout % 4:
        return None
    n = _lib.lib().mydet_wino_weights_floats(Cout, Cin)
    if n <= 0:
        return None
    u = torch.empty(n, dtype=torch.float32, device=w_ohwi.device)
    w = w_ohwi.contiguous()
    _lib.check(_lib.lib().mydet_wino_weights_f32(_ptr(w), Cout, Cin, _ptr(u), _stream()), 'mydet_wino_weights_f32')
    return u


# F(4x4,3x3) for the 3x3 layers with Cin >= WINO4_MIN_CIN whose grid fills the chip (>= WINO4_MIN_ITEMS workgroups of
# 32 tiles x 32 channels; smaller grids stay on F(2x2,3x3), which cuts them along K); MYDET_CONV_WINO4=0 turns it off.
# 512 = one full round of the resident workgroups (round 4; 768 before): the 16^2 layers of batch 32 at 512^2 are exactly
# that (2 134 -> 2 289 images/s), the FCOS head's dense 3x3 at 80^2 in a 16-image lane has 600 (+0.5 %)
WINOGRAD4 = os.environ.get('MYDET_CONV_WINO4', '1') != '0'
WINO4_MIN_CIN = int(os.environ.get('MYDET_WINO4_MIN_CIN', '64'))
WINO4_MIN_ITEMS = int(os.environ.get('MYDET_WINO4_MIN_ITEMS', '512'))


def wino4_items(B, H, W, Cout):
    """Workgroups of the F(4x4,3x3) kernel for one layer: 32 tiles of 4x4 outputs x 32 output channels each."""
    return -(-(B * -(-H // 4) * -(-W // 4)) // 32) * -(-Cout // 32)


def wino4_weights(w_ohwi):
    """Transform-domain copy of a 3x3 OHWI weight for the F(4x4,3x3) kernel (`conv2d(..., wino4=)`), or None when the
    shape is not covered (Cin % 4, Cout % 4)."""
    require_gpu(w_ohwi, 'wino4_weights')
    Cout, kh, kw, Cin = w_ohwi.shape
    if kh != 3 or kw != 3 or Cout % 4:
        return None
    n = _lib.lib().mydet_wino4_weights_floats(Cout, Cin)
    if n <= 0:
        return None
    u = torch.empty(n, dtype=torch.float32, device=w_ohwi.device)
    w = w_ohwi.contiguous()
    _lib.check(_lib.lib().mydet_wino4_weights_f32(_ptr(w), Cout, Cin, _ptr(u), _stream()), 'mydet_wino4_weights_f32')
    return u


# MYDET_CONV_SPLIT_BF16=0 keeps every direct conv on the float32 matrix instruction (A/B measurements)
SPLIT_BF16 = os.environ.get('MYDET_CONV_SPLIT_BF16', '1') != '0'


# The split-bf16 form takes a direct-conv layer from this many output pixels (B * Ho * Wo) up -- below, the float32 kernel's
# small-grid K cut is the tuned path (batch-1 layers) -- and, for 1x1 layers, from 128 output channels (tools/r05_b3.py:
# 128->64 @160^2 0.197 vs 0.187 ms for the float32 kernel; every wider shape of the headline 1.25-1.35 x faster)
B3_MIN_ROWS = int(os.environ.get('MYDET_B3_MIN_ROWS', '8192'))
B3_MIN_FLOP = float(os.environ.get('MYDET_B3_MIN_FLOP', '3e9'))
B3_KPAD = os.environ.get('MYDET_B3_KPAD', '1') != '0'      # 1x1 layers with Cin % 16 == 4, 8, 12 (last slab zero-filled)
# The EfficientNet expand convs (1x1, 6 x Cin output channels, swish) take it from fewer rows: on a batch lane of 8 / 16 images
# the 20^2 layers have 3 200 / 6 400 rows and still fill the chip (225-750 tiles of 128 x 128).  Measured in the model, two
# lanes (tools/r05_b3_effnet.sh, two runs each): expand convs on the float32 instruction 3 721 / 4 211 images/s (D1 batch 16 /
# D1-FCOS batch 32), split-bf16 from 8 192 rows 3 761 / 4 268, from 3 000 rows 3 859 / 4 350.  MYDET_B3_EXPAND_MIN_ROWS=0 = off.
B3_EXPAND_MIN_ROWS = int(os.environ.get('MYDET_B3_EXPAND_MIN_ROWS', '3000'))
# ... and the gated project convs (1x1, no activation, the squeeze-excite gate applied to the activations before they are split) from 64
# output channels and the same 3 000 rows.  Their grids are small (80-320 output channels: 50-200 tiles of 128 x 128 on a lane), so the
# launcher cuts a layer of at most cus / 2 such tiles into 64-row tiles.  Measured (tools/r05_b3_gate.sh, tools/r05_b3_half.sh; two to four
# runs each, one call): D1 batch 16 3 854 -> 3 893, D1-FCOS batch 32 4 407 -> 4 500 images/s; with 128-row tiles only, D1 LOST 1-2 %
# (3 851 -> 3 782) while D1-FCOS gained 1 %.  MYDET_B3_GATED_MIN_COUT=0 keeps them all on the float32 kernel.
B3_GATED_MIN_COUT = int(os.environ.get('MYDET_B3_GATED_MIN_COUT', '64'))
B3_GATED_MIN_ROWS = int(os.environ.get('MYDET_B3_GATED_MIN_ROWS', '3000'))


def b3_takes(M, Cin, Cout, k, min_rows=None, min_cout=128):
    """True when `conv2d(..., b3=)` runs the split-bf16 kernel for a layer of this shape."""
    if not SPLIT_BF16 or not (k > 1 or Cout >= min_cout):
        return False
    if Cin % 16 and not (B3_KPAD and k == 1 and Cin % 4 == 0):           # a 16-channel slab never straddles taps; a 1x1 layer's last slab may be short
        return False
    if min_rows is not None:
        return M >= min_rows
    # ... and from 3 GFLOP per launch: below, the float32 kernel's small tiles and K cut are the tuned path.  Batch 1 at 512^2 has two
    # layers past the row limit (32->64 and 64->128 stride 2: 2.4 GFLOP each on 512 / 128 tiles): 1.560 ms per image with them on
    # this kernel, 1.427 without (two runs each in one call); the smallest layers that gain at batch 32 have 3.4 GFLOP.
    return M >= B3_MIN_ROWS and 2.0 * M * k * k * Cin * Cout >= B3_MIN_FLOP


# conv_p3_kernel (csrc/conv_p3.hip: 3x3 conv with the workgroup's input patch resident in LDS, split-bf16 operands) takes a 3x3 pad-1
# layer when its 128-pixel tiles are mostly real pixels (P3_MIN_FILL; ragged 8 x 16 tiles waste their rows -- without strip tiles
# 256->512 @80->40 ran 0.65 vs 0.61 ms, 512->1024 @40->20 0.92 vs 0.69), the launch fills the chip (P3_MIN_WGS workgroups) and
#   stride 2: always (the first three stride-2 layers of Darknet-53 at 640^2: 1.07-1.11 x over conv_igemm_b3_kernel);
#   stride 1: only below WINO4_MIN_CIN input channels, where F(4x4) does not go (32->64 @320^2: 1.16-1.18 x over F(2x2)).
# profiles/r06_conv_p3.txt.  MYDET_CONV_P3=0 turns it off.
CONV_P3 = os.environ.get('MYDET_CONV_P3', '1') != '0'
P3_MIN_WGS = int(os.environ.get('MYDET_P3_MIN_WGS', '512'))             # a full round of the chip (two workgroups per CU); 512^2 batch 32: the 256->512
                                                                        # stride-2 layer @64->32 (1 024 workgroups) joins: 2 601-2 622 vs 2 597-2 601 images/s
P3_S1_MAX_CIN = int(os.environ.get('MYDET_P3_S1_MAX_CIN', '32'))      # stride-1 layers up to this many input channels
# tiles must be mostly real pixels: 0.75 admits the 40-wide (13 tiles for 1 600 pixels: 0.96) and 20-wide (4 for 400: 0.78) maps of Darknet-53 at
# 640^2 with their strip tiles, not a ragged 8 x 16 column on a 20-wide map (0.52).  MYDET_P3_STRIP=0: no strip tiles (the library reads it too)
P3_MIN_FILL = float(os.environ.get('MYDET_P3_MIN_FILL', '0.75'))
P3_STRIP = os.environ.get('MYDET_P3_STRIP', '1') != '0'


def p3_tiles(Ho, Wo, stride):
    """Workgroup tiles of 128 output pixels per image and channel tile that mydet_conv3x3_p3_f32 launches (csrc/conv_p3.hip): 8 x 16
    tiles; at stride 2 a remainder of 8 / 4 columns goes to 16 x 8 / 32 x 4 strip tiles, any other remainder to a ragged column."""
    rem = Wo % 16
    strip = stride == 2 and rem in (4, 8) and P3_STRIP
    tiles = (Wo // 16 if strip else -(-Wo // 16)) * -(-Ho // 8)
    if strip:
        tiles += -(-Ho // 16) if rem == 8 else -(-Ho // 32)
    return tiles


def p3_takes(B, Ho, Wo, Cin, Cout, k, stride, pad):
    """True when conv2d(..., b3=) hands the layer to conv3x3_p3."""
    if not (CONV_P3 and SPLIT_BF16) or k != 3 or stride not in (1, 2) or tuple(pad) != (1, 1, 1, 1) or Cin % 16:
        return False
    tiles = p3_tiles(Ho, Wo, stride)
    if B * tiles * -(-Cout // (128 if Cout > 64 else 64)) < P3_MIN_WGS or Ho * Wo < P3_MIN_FILL * 128 * tiles:
        return False
    return stride == 2 or Cin <= P3_S1_MAX_CIN


def split_bf16(w_ohwi):
    """The weight operand of `conv2d(..., b3=)`: the OHWI weight [Cout, kh, kw, Cin] as three bfloat16 planes, w = p0 + p1 + p2 to
    2^-27 |w|, in the split-bf16 kernels' slab-major order (include/mydet.h: mydet_split_bf16_f32); int16 storage.
    None when Cin % 16 (a 1x1 weight: when Cin % 4; its last 16-channel slab is zero-filled)."""
    require_gpu(w_ohwi, 'split_bf16')
    Cout = w_ohwi.shape[0]
    K = w_ohwi.numel() // Cout
    n = _lib.lib().mydet_split_bf16_elems(Cout, K)
    one_tap = w_ohwi.shape[1] == 1 and w_ohwi.shape[2] == 1
    if n <= 0 or (w_ohwi.shape[-1] % 16 and not (one_tap and w_ohwi.shape[-1] % 4 == 0)):
        return None
    w = w_ohwi.contiguous().float()
    out = torch.empty(n, dtype=torch.int16, device=w.device)
    _lib.check(_lib.lib().mydet_split_bf16_f32(_ptr(w), Cout, K, _ptr(out), _stream()), 'mydet_split_bf16_f32')
    return out


def conv2d(x, w_ohwi, scale, shift, k, stride, pad, act, residual=None, out=None, out_ld=None, gate=None, wino=None,
           wino4=None, interior=None, b3=None, b3_min_rows=None):
    """y = act(conv(x * gate)*scale + shift) + residual.  x logical [B,Cin,H,W]; pad=(top,left,bottom,right);
    gate: optional [B,Cin] per-image channel multipliers (squeeze-excite), 1x1 convs only;
    wino / wino4: optional `wino_weights(w_ohwi)` / `wino4_weights(w_ohwi)`: 3x3 stride-1 pad-1 layers then run a fused
    Winograd kernel -- F(4x4,3x3) when given and the grid fills the chip (or no F(2x2,3x3) weights are given).
    interior: 'in' / 'out' marks the input / output as a tensor that lives only inside a block (bookkeeping of the
    fused-minimum byte count of KernelTimer; no effect on the launch).
    b3: optional `split_bf16(w_ohwi)`: a layer that stays on the direct implicit GEMM then runs it on the bfloat16 matrix
    instructions with float32-exact split operands (include/mydet.h: mydet_conv2d_igemm_b3_f32; Cin % 16 == 0; with a gate: 1x1 layers
    without activation), from
    B3_MIN_ROWS output pixels up (`b3_min_rows` overrides)."""
    require_gpu(x, 'conv2d')
    if x.shape[1] % 4:
        raise ValueError(f'conv2d: Cin = {x.shape[1]} is not a multiple of 4 (the implicit-GEMM kernel reads channels in '
                         'float4; the 3-channel image layer goes through conv2d_stem)')
    x, ldx = to_nhwc(x)
    B, Cin, H, W = x.shape
    Cout = w_ohwi.shape[0]
    Ho = conv_out_size(H, k, stride, pad[0], pad[2])
    Wo = conv_out_size(W, k, stride, pad[1], pad[3])
    if out is None:
        out, ldy = empty_nhwc(B, Cout, Ho, Wo, x.device, out_ld)
    else:
        ldy = nhwc_ld(out)
        assert ldy is not None and out.shape == (B, Cout, Ho, Wo)
    ldr = 0
    if residual is not None:
        residual, ldr = to_nhwc(residual)
        assert residual.shape == out.shape
    if b3 is not None and gate is None and act in (ACT_NONE, ACT_LEAKY) and p3_takes(B, Ho, Wo, Cin, Cout, k, stride, pad):
        y = conv3x3_p3(x, b3, scale, shift, stride, act, residual=residual, out=out, cout=Cout)
        if y is not None:
            return y
    if (wino4 is not None and WINOGRAD and WINOGRAD4 and gate is None and k == 3 and stride == 1 and tuple(pad) == (1, 1, 1, 1)
            and ldy % 4 == 0 and ldr % 4 == 0 and (wino is None or wino4_items(B, H, W, Cout) >= WINO4_MIN_ITEMS)):
        ws = wino4_workspace(x.device, _lib.lib().mydet_wino4_workspace_bytes(B, H, W, Cin, Cout))
        t0 = TIMER.start() if TIMER else None
        code = _lib.lib().mydet_conv2d_wino4_f32(_ptr(x), ldx, _ptr(wino4), _ptr(scale), _ptr(shift), _ptr(residual), ldr,
                                                 _ptr(ws), ws.numel() * 4, _ptr(out), ldy, B, H, W, Cin, Cout, act, _stream())
        if t0:      # priced with the direct form's flops: the algorithmic work of the layer
            name = f'conv_wino4 {Cin}->{Cout} k3s1 {H}x{W}' if TIMER_DETAIL else 'conv_wino4'
            TIMER.stop(name, t0, 2.0 * B * Ho * Wo * Cout * 9 * Cin,
                       4.0 * (B * H * W * Cin + B * Ho * Wo * Cout * (2 if residual is not None else 1) + 9 * Cin * Cout))
        _lib.check(code, 'mydet_conv2d_wino4_f32')
        return out
    if (wino is not None and WINOGRAD and gate is None and k == 3 and stride == 1 and tuple(pad) == (1, 1, 1, 1)
            and ldy % 4 == 0 and ldr % 4 == 0):
        ws = conv_workspace(x.device)
        t0 = TIMER.start() if TIMER else None
        code = _lib.lib().mydet_conv2d_wino_f32(_ptr(x), ldx, _ptr(wino), _ptr(scale), _ptr(shift), _ptr(residual), ldr,
                                                _ptr(ws), ws.numel() * 4, _ptr(out), ldy, B, H, W, Cin, Cout, act,
                                                _stream())
        if t0:      # priced with the direct form's flops: the algorithmic work of the layer
            name = f'conv_wino {Cin}->{Cout} k3s1 {H}x{W}' if TIMER_DETAIL else 'conv_wino'
            TIMER.stop(name, t0, 2.0 * B * Ho * Wo * Cout * 9 * Cin,
                       4.0 * (B * H * W * Cin + B * Ho * Wo * Cout * (2 if residual is not None else 1) + 9 * Cin * Cout))
        _lib.check(code, 'mydet_conv2d_wino_f32')
        return out
    ws = conv_workspace(x.device)
    if (b3 is not None and b3_takes(B * Ho * Wo, Cin, Cout, k, b3_min_rows, min_cout=B3_GATED_MIN_COUT if gate is not None else 128)
            and (gate is None or (k == 1 and act == ACT_NONE))):
        t0 = TIMER.start() if TIMER else None
        code = _lib.lib().mydet_conv2d_igemm_b3_f32(
            _ptr(x), ldx, _ptr(b3), _ptr(scale), _ptr(shift), _ptr(residual), ldr, _ptr(gate), _ptr(ws), ws.numel() * 4, _ptr(out), ldy,
            B, H, W, Cin, Cout, k, k, stride, pad[0], pad[1], Ho, Wo, act, _stream())
        if code != -2:                              # MYDET_E_UNSUPP: the float32 kernel below
            if t0:
                name = f'conv_igemm_b3 {Cin}->{Cout} k{k}s{stride} {H}x{W}' if TIMER_DETAIL else 'conv_igemm_b3'
                b_in, b_out = 4.0 * B * H * W * Cin, 4.0 * B * Ho * Wo * Cout
                b_rest = 4.0 * (k * k * Cin * Cout) + (b_out if residual is not None else 0.0)
                TIMER.stop(name, t0, 2.0 * B * Ho * Wo * Cout * k * k * Cin, b_in + b_out + b_rest)
            _lib.check(code, 'mydet_conv2d_igemm_b3_f32')
            return out
    t0 = TIMER.start() if TIMER else None
    code = _lib.lib().mydet_conv2d_igemm_f32(
        _ptr(x), ldx, _ptr(w_ohwi), _ptr(scale), _ptr(shift), _ptr(residual), ldr, _ptr(gate),
        _ptr(ws), ws.numel() * 4 if ws is not None else 0, _ptr(out), ldy,
        B, H, W, Cin, Cout, k, k, stride, pad[0], pad[1], Ho, Wo, act, _stream())
    if t0:
        name = f'conv_igemm {Cin}->{Cout} k{k}s{stride} {H}x{W}' if TIMER_DETAIL else 'conv_igemm'
        b_in, b_out = 4.0 * B * H * W * Cin, 4.0 * B * Ho * Wo * Cout
        b_rest = 4.0 * (k * k * Cin * Cout) + (b_out if residual is not None else 0.0)
        TIMER.stop(name, t0, 2.0 * B * Ho * Wo * Cout * k * k * Cin, b_in + b_out + b_rest,
                   (0.0 if interior == 'in' else b_in) + (0.0 if interior == 'out' else b_out) + b_rest)
    _lib.check(code, 'mydet_conv2d_igemm_f32')
    return out


def conv3x3_p3(x, b3, scale, shift, stride, act, residual=None, out=None, out_ld=None, cout=None):
    """y = act(conv3x3_pad1(x) * scale + shift) + residual on the patch-resident split-bf16 kernel (csrc/conv_p3.hip,
    include/mydet.h: mydet_conv3x3_p3_f32).  b3 = split_bf16(w_ohwi) of the [Cout, 3, 3, Cin] weight; cout = Cout (default: len(shift)).
    Returns None when the kernel does not cover the shape (Cin % 16, activation): the caller then uses conv2d."""
    require_gpu(x, 'conv3x3_p3')
    x, ldx = to_nhwc(x)
    B, Cin, H, W = x.shape
    Cout = int(cout if cout is not None else (shift if shift is not None else scale).shape[0])
    if Cin % 16 or stride not in (1, 2) or act not in (ACT_NONE, ACT_LEAKY) or b3 is None:
        return None
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    if out is None:
        out, ldy = empty_nhwc(B, Cout, Ho, Wo, x.device, out_ld)
    else:
        ldy = nhwc_ld(out)
        assert ldy is not None and out.shape == (B, Cout, Ho, Wo)
    ldr = 0
    if residual is not None:
        residual, ldr = to_nhwc(residual)
        assert residual.shape == out.shape
    t0 = TIMER.start() if TIMER else None
    code = _lib.lib().mydet_conv3x3_p3_f32(_ptr(x), ldx, _ptr(b3), _ptr(scale), _ptr(shift), _ptr(residual), ldr, _ptr(out), ldy,
                                           B, H, W, Cin, Cout, stride, act, _stream())
    if code == -2:
        return None
    if t0:
        name = f'conv_p3 {Cin}->{Cout} k3s{stride} {H}x{W}' if TIMER_DETAIL else 'conv_p3'
        b_in, b_out = 4.0 * B * H * W * Cin, 4.0 * B * Ho * Wo * Cout
        TIMER.stop(name, t0, 2.0 * B * Ho * Wo * Cout * 9 * Cin, b_in + b_out + 4.0 * 9 * Cin * Cout + (b_out if residual is not None else 0.0))
    _lib.check(code, 'mydet_conv3x3_p3_f32')
    return out


def conv2d_stem(x, w_ohwi, scale, shift, stride, pad, act):
    """3->32 3x3 first layer; reads x with its own strides (NCHW or channels-last)."""
    require_gpu(x, 'conv2d_stem')
    assert x.dtype == torch.float32 and x.shape[1] == 3
    B, _, H, W = x.shape
    Cout = w_ohwi.shape[0]
    Ho = conv_out_size(H, 3, stride, pad[0], pad[2])
    Wo = conv_out_size(W, 3, stride, pad[1], pad[3])
    out, ldy = empty_nhwc(B, Cout, Ho, Wo, x.device)
    sb, sc, sh, sw = x.stride()
    t0 = TIMER.start() if TIMER else None
    code = _lib.lib().mydet_conv2d_stem_f32(_ptr(x), sb, sc, sh, sw, _ptr(w_ohwi), _ptr(scale), _ptr(shift),
                                            _ptr(out), ldy, B, H, W, Cout, stride, pad[0], pad[1], Ho, Wo, act,
                                            _stream())
    if t0:
        TIMER.stop('conv_stem', t0, *[4.0 * B * (3 * H * W + Cout * Ho * Wo)] * 2)        # bytes moved
    _lib.check(code, 'mydet_conv2d_stem_f32')
    return out


def se_slices(n_pixels):
    """Number of pixel slices the SE average is split over (deterministic two-stage sum)."""
    return max(1, min(128, n_pixels // 16))


def dwconv(x, w_kkc, scale, shift, k, stride, pad, act, squeeze=False, interior=False, se=None):
    """Depthwise k x k conv, y = act(conv*scale + shift); w_kkc [k,k,C]; pad=(top,left,bottom,right).
    squeeze=True also returns the per-slice channel sums [B,S,C] of y (input of `se_gate`).
    se=(w1 [Cse,C], b1 [Cse], w2t [Cse,C], b2 [C]): the launch also finishes the squeeze-excite gate of y
    (include/mydet.h: mydet_se_tail) and (y, gate [B,C]) is returned -- no squeeze sums, no gate launch."""
    require_gpu(x, 'dwconv')
    x, ldx = to_nhwc(x)
    B, C, H, W = x.shape
    Ho = conv_out_size(H, k, stride, pad[0], pad[2])
    Wo = conv_out_size(W, k, stride, pad[1], pad[3])
    out, ldy = empty_nhwc(B, C, Ho, Wo, x.device)
    partial, S = None, 0
    if squeeze or se is not None:
        S = _lib.lib().mydet_dwconv_slices(Ho, Wo, C, k, stride)      # what the kernel chosen for this layer writes
    if squeeze:
        partial = torch.empty((B, S + 1, C), dtype=torch.float32, device=x.device)    # slice S: scratch for the mean
    tail, gate, keep = _se_tail(se, B, C, _lib.lib().mydet_dwconv_se_groups(Ho, Wo, C, k, stride) if se is not None else 0, x.device)
    t0 = TIMER.start() if TIMER else None
    code = _lib.lib().mydet_dwconv_f32(_ptr(x), ldx, _ptr(w_kkc), _ptr(scale), _ptr(shift), _ptr(out), ldy, B, H, W, C,
                                       k, stride, pad[0], pad[1], Ho, Wo, act, _ptr(partial), S,
                                       ctypes.byref(tail) if tail is not None else None, _stream())
    if t0:
        TIMER.stop(f'dwconv {C} k{k}s{stride} {H}x{W}' if TIMER_DETAIL else 'dwconv', t0, *[4.0 * B * C * (H * W + Ho * Wo)] * 2,
                   fused=0.0 if interior is True else (4.0 * B * C * H * W if interior == 'out' else None))
    _lib.check(code, 'mydet_dwconv_f32')
    if se is not None:
        return out, gate
    return (out, partial) if squeeze else out


# (k, stride, Cin) the blocks use the fused launch for: the instantiated shapes where it beats expand + depthwise
# (tools/bench_mbconv.py, batch 16: 235 vs 386, 168 vs 243, 177 vs 192 us; (5,1,40) and (3,2,40) exist but lose: 145 vs 127, 95 vs 73)
MBCONV_FUSED_SHAPES = {(3, 2, 16), (3, 1, 24), (5, 2, 24)}
# MYDET_FUSED_MBCONV=0 keeps expand conv and depthwise conv as two launches (A/B measurements)
FUSED_MBCONV = os.environ.get('MYDET_FUSED_MBCONV', '1') != '0'


def fold_scale(w, scale):
    """Conv weight with the per-output-channel BatchNorm scale multiplied in (w OHWI [Cout,...] or depthwise [k,k,C])."""
    if w.dim() == 3:
        return (w * scale.view(1, 1, -1)).contiguous()
    return (w * scale.view(-1, *([1] * (w.dim() - 1)))).contiguous()


def mbconv_expand_dw(x, w_expand, shift0, w_dw, shift1, k, stride, pad, se=None):
    """swish(BN1(depthwise_k(swish(BN0(expand1x1(x)))))) and the per-tile channel sums of the result (SE squeeze) in
    one launch; w_expand OHWI [Cexp,1,1,Cin] and w_dw [k,k,Cexp] carry the BatchNorm scales (`fold_scale`), shift0 /
    shift1 are the folded shifts; pad=(top,left,bottom,right) of the expanded map.
    Returns (y [B,Cexp,Ho,Wo], partial [B,S+1,Cexp]) -- or, with se=(w1, b1, w2t, b2) as in `dwconv`, (y, gate [B,Cexp]):
    the launch finishes the squeeze-excite gate itself."""
    require_gpu(x, 'mbconv_expand_dw')
    x, ldx = to_nhwc(x)
    B, Cin, H, W = x.shape
    Cexp = w_expand.shape[0]
    Ho = conv_out_size(H, k, stride, pad[0], pad[2])
    Wo = conv_out_size(W, k, stride, pad[1], pad[3])
    out, ldy = empty_nhwc(B, Cexp, Ho, Wo, x.device)
    S = _lib.lib().mydet_mbconv_tiles(Ho, Wo, stride)
    partial = torch.empty((B, S + 1, Cexp), dtype=torch.float32, device=x.device) if se is None else None
    tail, gate, keep = _se_tail(se, B, Cexp, S, x.device)
    t0 = TIMER.start() if TIMER else None
    code = _lib.lib().mydet_mbconv_expand_dw_f32(_ptr(x), ldx, _ptr(w_expand), _ptr(shift0), _ptr(w_dw),
                                                 _ptr(shift1), _ptr(out), ldy, B, H, W, Cin, Cexp, k, stride,
                                                 pad[0], pad[1], Ho, Wo, _ptr(partial), S,
                                                 ctypes.byref(tail) if tail is not None else None, _stream())
    if t0:      # algorithmic bytes of the two reference layers it replaces: expand (in + out) and depthwise (in + out)
        nb = 4.0 * B * (H * W * (Cin + Cexp) + Cexp * (H * W + Ho * Wo))
        TIMER.stop(f'mbconv_expand_dw {Cin}->{Cexp} k{k}s{stride} {H}x{W}' if TIMER_DETAIL else 'mbconv_expand_dw', t0,
                   2.0 * B * H * W * Cin * Cexp, nb, fused=4.0 * B * H * W * Cin)
    _lib.check(code, 'mydet_mbconv_expand_dw_f32')
    return (out, gate) if se is not None else (out, partial)


# MYDET_FUSED_STEM=0 keeps the EfficientNet stem and block 0's depthwise conv as two launches
FUSED_STEM_DW = os.environ.get('MYDET_FUSED_STEM', '1') != '0'


def stem_dw(x, w_stem, shift0, w_dw, shift1, pad, se=None):
    """swish(BN1(depthwise3x3(swish(BN0(conv3x3_s2(image)))))) and the per-tile channel sums of the result in one launch
    (EfficientNet stem + block 0's depthwise conv).  x [B,3,H,W] in any strides; w_stem OHWI [32,3,3,3] and w_dw
    [3,3,32] carry the BatchNorm scales (`fold_scale`); pad = the stem's (top, left, bottom, right).
    Returns (y [B,32,Hs,Ws], partial [B,S+1,32]) -- or, with se=(w1, b1, w2t, b2) as in `dwconv`, (y, gate [B,32])."""
    require_gpu(x, 'stem_dw')
    assert x.dtype == torch.float32 and x.shape[1] == 3
    B, _, H, W = x.shape
    C = w_stem.shape[0]
    Hs = conv_out_size(H, 3, 2, pad[0], pad[2])
    Ws = conv_out_size(W, 3, 2, pad[1], pad[3])
    out, ldy = empty_nhwc(B, C, Hs, Ws, x.device)
    S = _lib.lib().mydet_mbconv_tiles(Hs, Ws, 1)
    partial = torch.empty((B, S + 1, C), dtype=torch.float32, device=x.device) if se is None else None
    tail, gate, keep = _se_tail(se, B, C, S, x.device)
    sb, sc, sh, sw = x.stride()
    t0 = TIMER.start() if TIMER else None
    code = _lib.lib().mydet_stem_dw_f32(_ptr(x), sb, sc, sh, sw, _ptr(w_stem), _ptr(shift0), _ptr(w_dw), _ptr(shift1), _ptr(out),
                                        ldy, B, H, W, C, pad[0], pad[1], Hs, Ws, _ptr(partial), S,
                                        ctypes.byref(tail) if tail is not None else None, _stream())
    if t0:      # reference-layer bytes: stem (image in, map out) + depthwise (map in, map out)
        TIMER.stop('stem_dw', t0, 0.0, 4.0 * B * (3 * H * W + 3 * C * Hs * Ws), fused=4.0 * B * (3 * H * W + C * Hs * Ws))
    _lib.check(code, 'mydet_stem_dw_f32')
    return (out, gate) if se is not None else (out, partial)


def channel_sums(x):
    """Per-slice channel sums [B,S,C] of x [B,C,H,W] (standalone squeeze)."""
    require_gpu(x, 'channel_sums')
    x, ldx = to_nhwc(x)
    B, C, H, W = x.shape
    S = se_slices(H * W)
    partial = torch.empty((B, S + 1, C), dtype=torch.float32, device=x.device)
    code = _lib.lib().mydet_channel_sums_f32(_ptr(x), ldx, B, H, W, C, _ptr(partial), S, _stream())
    _lib.check(code, 'mydet_channel_sums_f32')
    return partial


def se_gate(partial, n_pixels, w1, b1, w2t, b2):
    """gate [B,C] = sigmoid(W2 . swish(W1 . mean + b1) + b2) from per-slice sums [B,S+1,C]; w2t = W2 transposed [Cse,C]."""
    require_gpu(partial, 'se_gate')
    B, S, C = partial.shape
    S -= 1
    Cse = w1.shape[0]
    gate = torch.empty((B, C), dtype=torch.float32, device=partial.device)
    t0 = TIMER.start() if TIMER else None
    code = _lib.lib().mydet_se_gate_f32(_ptr(partial), S, B, n_pixels, C, _ptr(w1), _ptr(b1), Cse, _ptr(w2t), _ptr(b2),
                                        _ptr(gate), _stream())
    if t0:
        TIMER.stop('se_gate', t0, *[4.0 * B * S * C] * 2, fused=0.0)
    _lib.check(code, 'mydet_se_gate_f32')
    return gate


def maxpool3s2(x):
    """max_pool2d(x, kernel_size=3, stride=2, padding=1)."""
    require_gpu(x, 'maxpool3s2')
    x, ldx = to_nhwc(x)
    B, C, H, W = x.shape
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    out, ldy = empty_nhwc(B, C, Ho, Wo, x.device)
    t0 = TIMER.start() if TIMER else None
    code = _lib.lib().mydet_maxpool3s2_f32(_ptr(x), ldx, _ptr(out), ldy, B, H, W, C, Ho, Wo, _stream())
    if t0:
        TIMER.stop('maxpool', t0, *[4.0 * B * C * (H * W + Ho * Wo)] * 2)
    _lib.check(code, 'mydet_maxpool3s2_f32')
    return out


FUSE_SAME, FUSE_UP2X, FUSE_POOL = 0, 1, 2


def bifpn_fuse(inputs, modes, weights):
    """swish(sum_i w_i * in_i) with w = relu(weights)/(sum+1e-4).  inputs: 2-3 tensors; modes[i] in
    FUSE_SAME / FUSE_UP2X (half-size map, nearest 2x) / FUSE_POOL (double-size map, max-pool 3/2/1).
    The output has the size of the FUSE_SAME input(s)."""
    n = len(inputs)
    require_gpu(inputs[0], 'bifpn_fuse')
    prepared = [to_nhwc(t) for t in inputs]
    ref = next(t for (t, _), m in zip(prepared, modes) if m == FUSE_SAME)
    B, C, H, W = ref.shape
    for (t, _), m in zip(prepared, modes):
        exp = {FUSE_SAME: (H, W), FUSE_UP2X: (H // 2, W // 2), FUSE_POOL: (H * 2, W * 2)}[m]
        assert tuple(t.shape) == (B, C) + exp, (tuple(t.shape), m, (B, C, H, W))
    out, ldy = empty_nhwc(B, C, H, W, ref.device)
    args = []
    for i in range(3):
        if i < n:
            args += [_ptr(prepared[i][0]), prepared[i][1], int(modes[i])]
        else:
            args += [ctypes.c_void_p(0), 0, 0]
    t0 = TIMER.start() if TIMER else None
    code = _lib.lib().mydet_bifpn_fuse_f32(n, *args, _ptr(weights), _ptr(out), ldy, B, H, W, C, _stream())
    if t0:
        TIMER.stop('bifpn_fuse', t0, *[4.0 * B * C * H * W * (n + 1)] * 2)
    _lib.check(code, 'mydet_bifpn_fuse_f32')
    return out


def pack_pointwise(w):
    """Pointwise weight [Cout, C] (or OHWI [Cout,1,1,C]) in the MFMA operand order `sepconv_nodes` reads, four k-steps of
    a lane side by side (one 16-byte load / LDS read per four MFMAs; include/mydet.h):
        packed[nb][kq][lane][e] = W[16*nb + (lane & 15)][4*(4*kq + e) + (lane >> 4)],   kq < ceil(C/16),
    zero for rows >= Cout and k >= C."""
    w = w.reshape(w.shape[0], -1).float()
    Cout, C = w.shape
    assert C % 4 == 0
    nb, ks = (Cout + 15) // 16, C // 4
    kq = (ks + 3) // 4
    wp = w.new_zeros((nb * 16, kq * 4, 4))                 # [row][k-step][k within the step]
    wp[:Cout, :ks] = w.view(Cout, ks, 4)
    # [nb][16 rows][kq][e][4 kk] -> [nb][kq][kk][16 rows][e]: lane = kk * 16 + row
    return wp.view(nb, 16, kq, 4, 4).permute(0, 2, 4, 1, 3).contiguous().view(nb, kq, 64, 4)


SEPCONV_CHANNELS = (88,)        # instantiated channel counts of the fused node kernel
# MYDET_FUSED_NODES=0 keeps the pyramid on the three-launch path (fusion, depthwise, pointwise GEMM): A/B measurements
FUSED_NODES = os.environ.get('MYDET_FUSED_NODES', '1') != '0'


def sepconv_nodes(nodes):
    """Fused [fusion + swish ->] depthwise 3x3 -> pointwise 1x1 (+ folded BN, act) nodes, up to 10 per launch.
    nodes: list of dicts with keys inputs (1-3 tensors [B,C,h,w]), modes (FUSE_*), fuse_weights (raw, for >= 2 inputs),
    w_dw [3,3,C], w_pw (pack_pointwise), scale (or None), shift, cout, act, and optionally out (a tensor to write).
    Returns the output tensors."""
    assert 1 <= len(nodes) <= _lib.SEPCONV_MAX_NODES
    arr = (_lib.SepconvNode * len(nodes))()
    outs, keep = [], []
    B = C = None
    work = fused = 0.0
    for i, nd in enumerate(nodes):
        prepared = [to_nhwc(t) for t in nd['inputs']]
        require_gpu(prepared[0][0], 'sepconv_nodes')
        modes = list(nd.get('modes') or [FUSE_SAME] * len(prepared))
        ref = next(t for (t, _), m in zip(prepared, modes) if m == FUSE_SAME)
        b, c, H, W = ref.shape
        B, C = (b, c) if B is None else (B, C)
        assert (b, c) == (B, C), 'all nodes of a launch share batch and channels'
        for (t, _), m in zip(prepared, modes):
            exp = {FUSE_SAME: (H, W), FUSE_UP2X: (H // 2, W // 2), FUSE_POOL: (H * 2, W * 2)}[m]
            assert tuple(t.shape) == (B, C) + exp, (tuple(t.shape), m, (B, C, H, W))
        cout = nd['cout']
        out = nd.get('out')
        if out is None:
            out, ldy = empty_nhwc(B, cout, H, W, ref.device)
        else:
            ldy = nhwc_ld(out)
            assert ldy is not None and tuple(out.shape) == (B, cout, H, W)
        n_in = len(prepared)
        node = arr[i]
        for k in range(3):
            node.inp[k] = prepared[k][0].data_ptr() if k < n_in else None
            node.ld[k] = prepared[k][1] if k < n_in else 0
            node.mode[k] = int(modes[k]) if k < n_in else 0
        node.n_in = n_in
        fw = nd.get('fuse_weights')
        node.fuse_weights = fw.data_ptr() if n_in > 1 else None
        node.w_dw, node.w_pw_packed = nd['w_dw'].data_ptr(), nd['w_pw'].data_ptr()
        node.scale = nd['scale'].data_ptr() if nd.get('scale') is not None else None
        node.shift = nd['shift'].data_ptr()
        node.y, node.ldy, node.H, node.W, node.Cout, node.act = out.data_ptr(), ldy, H, W, cout, int(nd.get('act', ACT_NONE))
        keep.append((prepared, fw, out))
        outs.append(out)
        # algorithmic bytes of the reference layers this node replaces: fusion (n+1 maps), depthwise (2), pointwise (C + Cout)
        px = 4.0 * B * H * W
        work += px * ((C * (n_in + 1) if n_in > 1 else 0) + 2 * C + C + cout)
        fused += sum(4.0 * t.numel() for t, _ in prepared) + px * cout
    t0 = TIMER.start() if TIMER else None
    code = _lib.lib().mydet_sepconv_nodes_f32(len(nodes), ctypes.cast(arr, ctypes.c_void_p), B, C, _stream())
    if t0:
        TIMER.stop('sepconv_nodes', t0, work, work, fused=fused)
    _lib.check(code, 'mydet_sepconv_nodes_f32')
    return outs


# MYDET_FUSED_DECODE=0: the EfDetHead + RetinaLayer path writes its class logits and decodes them in a second launch
FUSED_DECODE = os.environ.get('MYDET_FUSED_DECODE', '1') != '0'


def pack_pointwise_per_anchor(w, shift, A, n_cls):
    """Class-tower weights [A * n_cls, C] / shifts [A * n_cls] with every anchor's rows padded to whole 16-channel blocks
    (zero rows / zero shifts), in `pack_pointwise` order: what `sepconv_decode_retina` reads (include/mydet.h)."""
    w = w.reshape(w.shape[0], -1).float()
    C = w.shape[1]
    cpad = (n_cls + 15) // 16 * 16
    wp = w.new_zeros((A, cpad, C))
    wp[:, :n_cls] = w.view(A, n_cls, C)
    sp = shift.new_zeros((A, cpad))
    sp[:, :n_cls] = shift.float().view(A, n_cls)
    return pack_pointwise(wp.view(A * cpad, C)), sp.view(-1).contiguous()


def sepconv_decode_retina(nodes, A, n_cls, img_hw, bbox, class_idx, score):
    """The last sepconv of every EfDetHead tower with RetinaLayer's decode in its epilogue, one launch.
    nodes: dicts with keys inputs ([tensor [B,C,h,w]]), w_dw, w_pw, shift, scale (or None), kind (0 class tower: w_pw /
    shift from `pack_pointwise_per_anchor`; 1 box tower: plain `pack_pointwise`), stride, anchors_wh ([A,2], box
    towers), n_off.  Writes bbox [B,N,4], class_idx [B,N] i64, score [B,N] at the nodes' candidate ranges."""
    assert 1 <= len(nodes) <= _lib.SEPCONV_MAX_NODES
    require_gpu(bbox, 'sepconv_decode_retina')
    arr = (_lib.SepconvDecodeNode * len(nodes))()
    keep = []
    B = C = None
    work = fused = 0.0
    cpad = (n_cls + 15) // 16 * 16
    for i, nd in enumerate(nodes):
        x, ld = to_nhwc(nd['inputs'][0])
        b, c, H, W = x.shape
        B, C = (b, c) if B is None else (B, C)
        assert (b, c) == (B, C), 'all nodes of a launch share batch and channels'
        node = arr[i].node
        for k in range(3):
            node.inp[k] = x.data_ptr() if k == 0 else None
            node.ld[k] = ld if k == 0 else 0
            node.mode[k] = 0
        node.n_in = 1
        node.fuse_weights = None
        node.w_dw, node.w_pw_packed = nd['w_dw'].data_ptr(), nd['w_pw'].data_ptr()
        node.scale = nd['scale'].data_ptr() if nd.get('scale') is not None else None
        node.shift = nd['shift'].data_ptr()
        node.y, node.ldy, node.H, node.W, node.act = None, 0, H, W, ACT_NONE
        node.Cout = A * cpad if nd['kind'] == 0 else A * 4
        arr[i].kind, arr[i].stride, arr[i].n_off = int(nd['kind']), float(nd['stride']), int(nd['n_off'])
        anch = None
        if nd['kind'] == 1:
            anch = np.ascontiguousarray(np.asarray(nd['anchors_wh'], dtype=np.float32).reshape(-1))
            assert anch.size == 2 * A
        arr[i].anchors_wh = anch.ctypes.data if anch is not None else None
        keep.append((x, anch))
        # reference-layer bytes: depthwise (2 maps), pointwise (C in, Cout out), decode (Cout in, 28 B per candidate out)
        px = 4.0 * B * H * W
        cout = A * n_cls if nd['kind'] == 0 else A * 4
        work += px * (2 * C + C + cout + cout) + (12.0 if nd['kind'] == 0 else 16.0) * B * A * H * W
        fused += px * C + (12.0 if nd['kind'] == 0 else 16.0) * B * A * H * W
    t0 = TIMER.start() if TIMER else None
    code = _lib.lib().mydet_sepconv_decode_retina_f32(len(nodes), ctypes.cast(arr, ctypes.c_void_p), B, C, A, n_cls,
                                                      int(img_hw[0]), int(img_hw[1]), _ptr(bbox), _ptr(class_idx),
                                                      _ptr(score), bbox.shape[1], _stream())
    if t0:
        TIMER.stop('sepconv_decode', t0, work, work, fused=fused)
    _lib.check(code, 'mydet_sepconv_decode_retina_f32')


def upsample_concat(a, size, b=None):
    """cat((nearest_resize(a, size), b), dim=1) in one pass."""
    require_gpu(a, 'upsample_concat')
    a, lda = to_nhwc(a)
    B, C1, Ha, Wa = a.shape
    Ho, Wo = size
    C2, ldb = 0, 0
    if b is not None:
        b, ldb = to_nhwc(b)
        C2 = b.shape[1]
        assert b.shape[0] == B and tuple(b.shape[2:]) == (Ho, Wo)
    out, ldy = empty_nhwc(B, C1 + C2, Ho, Wo, a.device)
    t0 = TIMER.start() if TIMER else None
    code = _lib.lib().mydet_upsample_concat_f32(_ptr(a), lda, Ha, Wa, C1, _ptr(b), ldb, C2, _ptr(out), ldy, B, Ho,
                                                Wo, _stream())
    if t0:
        TIMER.stop('upsample_concat', t0, *[4.0 * B * (C1 * Ha * Wa + C2 * Ho * Wo + (C1 + C2) * Ho * Wo)] * 2)
    _lib.check(code, 'mydet_upsample_concat_f32')
    return out


# MYDET_FUSED_UPCAT=0 keeps nearest-upsample + concat and the 1x1 conv behind them as two launches (A/B measurements)
FUSED_UPCAT = os.environ.get('MYDET_FUSED_UPCAT', '1') != '0'


def conv1x1_upcat(a, b, w_ohwi, scale, shift, act):
    """act(conv1x1(cat((nearest_2x(a), b), dim=1)) * scale + shift) in ONE launch -- the concatenated tensor is read on the
    fly, never written; bit-identical to `upsample_concat` + `conv2d` for every shape it accepts.  a [B,C1,H/2,W/2],
    b [B,C2,H,W].  Returns None when the shape is not covered -- channel counts not multiples of 32, or a shape `conv2d` would
    not run on the 64 x 64 x 32 tile the fused launch is instantiated for (include/mydet.h) -- and the caller then runs the two
    launches."""
    require_gpu(a, 'conv1x1_upcat')
    B, C1, Ha, Wa = a.shape
    _, C2, H, W = b.shape
    Cout = w_ohwi.shape[0]
    if not FUSED_UPCAT or (H, W) != (2 * Ha, 2 * Wa) or C1 % 32 or C2 % 32 or act != ACT_LEAKY or w_ohwi.shape[1:3] != (1, 1):
        return None
    a, lda = to_nhwc(a)
    b, ldb = to_nhwc(b)
    out, ldy = empty_nhwc(B, Cout, H, W, a.device)
    ws = conv_workspace(a.device)
    t0 = TIMER.start() if TIMER else None
    code = _lib.lib().mydet_conv1x1_upcat_f32(_ptr(a), lda, C1, _ptr(b), ldb, C2, _ptr(w_ohwi), _ptr(scale), _ptr(shift), _ptr(ws),
                                              ws.numel() * 4, _ptr(out), ldy, B, H, W, Cout, act, _stream())
    if code == -2:                                  # MYDET_E_UNSUPP: the caller runs the two launches
        return None
    if t0:
        name = f'conv_igemm {C1}^+{C2}->{Cout} k1s1 {H}x{W}' if TIMER_DETAIL else 'conv_igemm'
        TIMER.stop(name, t0, 2.0 * B * H * W * Cout * (C1 + C2), 4.0 * (B * (Ha * Wa * C1 + H * W * C2) + B * H * W * Cout + (C1 + C2) * Cout))
    _lib.check(code, 'mydet_conv1x1_upcat_f32')
    return out


def space_to_depth(x):
    """Focus' 2x2 space-to-depth (external/ultralytics/common.py:84-86): [B,C,H,W] (any strides) -> [B,4C,H/2,W/2]
    channels-last, channel g*C + c with g = 0:(dy 0,dx 0) 1:(1,0) 2:(0,1) 3:(1,1)."""
    require_gpu(x, 'space_to_depth')
    assert x.dtype == torch.float32 and x.dim() == 4
    B, C, H, W = x.shape
    assert H % 2 == 0 and W % 2 == 0
    out, ldy = empty_nhwc(B, 4 * C, H // 2, W // 2, x.device)
    sb, sc, sh, sw = x.stride()
    t0 = TIMER.start() if TIMER else None
    code = _lib.lib().mydet_space_to_depth_f32(_ptr(x), sb, sc, sh, sw, _ptr(out), ldy, B, C, H, W, _stream())
    if t0:
        TIMER.stop('space_to_depth', t0, *[8.0 * B * C * H * W] * 2)
    _lib.check(code, 'mydet_space_to_depth_f32')
    return out


def spp_concat(x, ks=(5, 9, 13)):
    """cat([x] + [max_pool2d(x, k, 1, k // 2) for k in ks], 1) in one pass (external/ultralytics/common.py:68-70)."""
    require_gpu(x, 'spp_concat')
    assert len(ks) == 3
    x, ldx = to_nhwc(x)
    B, C, H, W = x.shape
    assert C % 4 == 0
    out, ldy = empty_nhwc(B, 4 * C, H, W, x.device)
    t0 = TIMER.start() if TIMER else None
    code = _lib.lib().mydet_spp_concat_f32(_ptr(x), ldx, _ptr(out), ldy, B, H, W, C, int(ks[0]), int(ks[1]), int(ks[2]), _stream())
    if t0:
        TIMER.stop('spp_concat', t0, *[4.0 * B * H * W * C * 5] * 2)
    _lib.check(code, 'mydet_spp_concat_f32')
    return out


def decode(mode, box, ldbox, box_astride, box_c0, cls, ldcls, cls_astride, cls_c0, conf_c0, anchors_wh, A, C,
           B, H, W, stride, img_hw, bbox, class_idx, score, n_off):
    """Decode one level into bbox[B,N,4] / class_idx[B,N] / score[B,N] at candidate offset n_off."""
    require_gpu(box, 'decode')
    N = bbox.shape[1]
    anch = None
    if anchors_wh is not None:
        anch = np.ascontiguousarray(np.asarray(anchors_wh, dtype=np.float32).reshape(-1))
        assert anch.size == 2 * A
    t0 = TIMER.start() if TIMER else None
    code = _lib.lib().mydet_decode_f32(
        mode, _ptr(box), ldbox, box_astride, box_c0, _ptr(cls), ldcls, cls_astride, cls_c0, conf_c0,
        ctypes.c_void_p(anch.ctypes.data) if anch is not None else ctypes.c_void_p(0), A, C, B, H, W,
        float(stride), int(img_hw[0]), int(img_hw[1]), _ptr(bbox), _ptr(class_idx), _ptr(score), N, n_off,
        _stream())
    if t0:      # algorithmic bytes: every head logit once + 28 B per candidate
        per_pix = A * (C + 4 + (0 if mode == DECODE_RETINA else 1))
        TIMER.stop('decode', t0, *[4.0 * B * H * W * per_pix + 28.0 * B * A * H * W] * 2)
    _lib.check(code, 'mydet_decode_f32')


def decode_levels(mode, levels, box_astride, box_c0, cls_astride, cls_c0, conf_c0, A, C, B, img_hw, bbox, class_idx,
                  score):
    """Decode every pyramid level with one launch.  levels: list of dicts with keys box, ldbox, cls, ldcls,
    anchors_wh (array-like [A,2] or None), H, W, stride, n_off."""
    require_gpu(bbox, 'decode_levels')
    n = len(levels)
    arr = (_lib.DecodeLevel * n)()
    keep = []                                         # host anchor arrays must outlive the call
    work = 0.0
    for i, lv in enumerate(levels):
        anch = None
        if lv['anchors_wh'] is not None:
            anch = np.ascontiguousarray(np.asarray(lv['anchors_wh'], dtype=np.float32).reshape(-1))
            assert anch.size == 2 * A
            keep.append(anch)
        arr[i] = _lib.DecodeLevel(lv['box'].data_ptr(), lv['ldbox'], lv['cls'].data_ptr(), lv['ldcls'],
                                  anch.ctypes.data if anch is not None else None, lv['H'], lv['W'], float(lv['stride']),
                                  lv['n_off'])
        per_pix = A * (C + 4 + (0 if mode == DECODE_RETINA else 1))
        work += 4.0 * B * lv['H'] * lv['W'] * per_pix + 28.0 * B * A * lv['H'] * lv['W']
    t0 = TIMER.start() if TIMER else None
    code = _lib.lib().mydet_decode_levels_f32(mode, n, ctypes.cast(arr, ctypes.c_void_p), box_astride, box_c0,
                                              cls_astride, cls_c0, conf_c0, A, C, B, int(img_hw[0]), int(img_hw[1]),
                                              _ptr(bbox), _ptr(class_idx), _ptr(score), bbox.shape[1], _stream())
    if t0:
        TIMER.stop('decode', t0, work, work)
    _lib.check(code, 'mydet_decode_levels_f32')


def record_views(records):
    """Field views of a detection-record buffer (int32 [B, REC_WORDS], include/mydet.h MYDET_REC_*): nothing is
    copied -- the dict the rest of the package works with IS the wire buffer of the multi-GPU exchange."""
    B = records.shape[0]
    assert records.dtype == torch.int32 and records.shape[1] == _lib.REC_WORDS and records.is_contiguous()
    k = _lib.REC_TOPK
    return {'count': records[:, _lib.REC_COUNT],
            'bbox': records[:, _lib.REC_BBOX:_lib.REC_SCORE].view(torch.float32).view(B, k, 4),
            'score': records[:, _lib.REC_SCORE:_lib.REC_CLASS].view(torch.float32),
            'class_idx': records[:, _lib.REC_CLASS:_lib.REC_INDEX].view(torch.int64),
            'index': records[:, _lib.REC_INDEX:_lib.REC_WORDS],
            'records': records}


def postprocess(bbox, class_idx, score, conf_thres, nms_thres, topk=TOPK, records=None):
    """Batched filter/top-k/class-aware NMS.  bbox [B,N,4], class_idx [B,N] i64, score [B,N].

    Returns dict of device tensors: count [B] i32, bbox [B,512,4], class_idx [B,512] i64, score [B,512],
    index [B,512] i32 -- all views of 'records' (int32 [B, REC_WORDS]), which the kernel writes directly in the
    wire layout of the all-gather (parallel.gather_detections).  Class ids must lie in [0, 4096): an image whose
    selected candidates break that gets count = -1 (`check_counts` raises on it) instead of aliased classes.
    """
    require_gpu(bbox, 'postprocess')
    assert bbox.dtype == torch.float32 and score.dtype == torch.float32 and class_idx.dtype == torch.int64
    assert topk == TOPK
    bbox, class_idx, score = bbox.contiguous(), class_idx.contiguous(), score.contiguous()
    B, N = score.shape
    dev = bbox.device
    if records is None:
        records = torch.empty((B, _lib.REC_WORDS), dtype=torch.int32, device=dev)
    assert records.dtype == torch.int32 and tuple(records.shape) == (B, _lib.REC_WORDS) and records.is_contiguous()
    scratch = torch.empty((B, max(N, 1)), dtype=torch.int64, device=dev)
    t0 = TIMER.start() if TIMER else None
    code = _lib.lib().mydet_postprocess_records_f32(_ptr(bbox), _ptr(class_idx), _ptr(score), B, N, float(conf_thres),
                                                    float(nms_thres), _ptr(records), _ptr(scratch), _stream())
    if t0:
        TIMER.stop('postprocess', t0, float(B))
    _lib.check(code, 'mydet_postprocess_records_f32')
    return record_views(records)


def check_counts(counts):
    """Raise on the kernel's error sentinel in a host copy of the per-image counts (include/mydet.h:
    MYDET_COUNT_BAD_CLASS)."""
    bad = [b for b, k in enumerate(counts) if k < 0]
    if bad:
        raise ValueError(f'postprocess: class ids outside [0, 4096) among the selected candidates of image(s) {bad}; '
                         'the class-aware NMS key holds 12 class bits')
    return counts


def bboxes_iou(a, b, xyxy=False):
    require_gpu(a, 'bboxes_iou')
    a, b = a.contiguous().float(), b.contiguous().float()
    out = torch.empty((a.shape[0], b.shape[0]), dtype=torch.float32, device=a.device)
    code = _lib.lib().mydet_bboxes_iou_f32(_ptr(a), a.shape[0], _ptr(b), b.shape[0], 1 if xyxy else 0, _ptr(out),
                                           _stream())
    _lib.check(code, 'mydet_bboxes_iou_f32')
    return out


def cxcywh_to_x1y1x2y2(boxes):
    """Rows (cx, cy, w, h[, ...]) -> (x1, y1, x2, y2[, ...]) in a new tensor of the same shape (include/mydet.h:
    mydet_cxcywh_to_x1y1x2y2_f32); any leading dimensions, last dimension >= 4."""
    require_gpu(boxes, 'cxcywh_to_x1y1x2y2')
    if boxes.dtype != torch.float32:
        raise TypeError(f'cxcywh_to_x1y1x2y2: float32 boxes expected, got {boxes.dtype}')
    src = boxes.contiguous()
    out = torch.empty_like(src)
    width = src.shape[-1]
    code = _lib.lib().mydet_cxcywh_to_x1y1x2y2_f32(_ptr(src), _ptr(out), src.numel() // width if width else 0, width, _stream())
    _lib.check(code, 'mydet_cxcywh_to_x1y1x2y2_f32')
    return out


def bboxes_to_original_(bbox, pad_info):
    require_gpu(bbox, 'bboxes_to_original_')
    assert bbox.is_contiguous() and bbox.dtype == torch.float32 and bbox.shape[-1] == 4
    ori_w, ori_h, tl_x, tl_y, imw, imh = [float(v) for v in pad_info]
    code = _lib.lib().mydet_bboxes_to_original_f32(_ptr(bbox), bbox.shape[0], ori_w, ori_h, tl_x, tl_y, imw, imh,
                                                   _stream())
    _lib.check(code, 'mydet_bboxes_to_original_f32')
    return bbox


def resize_bilinear_u8(src_u8, out_hw, dst=None, top=0, left=0):
    """PIL-exact bilinear resize of one uint8 image [H,W,3] on the device (include/mydet.h: mydet_resize_bilinear_u8).
    dst: optional uint8 [Hd,Wd,3] to write into at (top, left) -- e.g. one image of a zero-padded batch buffer."""
    from .utils.image_ops import resample_tables
    require_gpu(src_u8, 'resize_bilinear_u8')
    assert src_u8.dtype == torch.uint8 and src_u8.dim() == 3 and src_u8.shape[2] == 3 and src_u8.is_contiguous()
    H, W, _ = src_u8.shape
    oh, ow = int(out_hw[0]), int(out_hw[1])
    if dst is None:
        dst = torch.empty((oh, ow, 3), dtype=torch.uint8, device=src_u8.device)
    assert dst.dtype == torch.uint8 and dst.dim() == 3 and dst.shape[2] == 3 and dst.stride(2) == 1 and dst.stride(1) == 3
    assert top + oh <= dst.shape[0] and left + ow <= dst.shape[1]
    tabs = []
    for n_in, n_out in ((W, ow), (H, oh)):
        if n_in == n_out:
            tabs.append((None, None, 0))
        else:
            key = (n_in, n_out, str(src_u8.device))
            hit = _RESAMPLE_CACHE.get(key)
            if hit is None:
                b, k = resample_tables(n_in, n_out)
                hit = (torch.from_numpy(b).to(src_u8.device), torch.from_numpy(k).to(src_u8.device), k.shape[1])
                if len(_RESAMPLE_CACHE) > 256:
                    _RESAMPLE_CACHE.clear()
                _RESAMPLE_CACHE[key] = hit
            tabs.append(hit)
    (bx, kx, ksx), (by, ky, ksy) = tabs
    dptr = ctypes.c_void_p(dst.data_ptr() + top * dst.stride(0) + left * 3)
    code = _lib.lib().mydet_resize_bilinear_u8(_ptr(src_u8), H, W, W * 3, dptr, oh, ow, dst.stride(0), _ptr(bx), _ptr(kx), ksx,
                                               _ptr(by), _ptr(ky), ksy, _stream())
    _lib.check(code, 'mydet_resize_bilinear_u8')
    return dst


_RESAMPLE_CACHE = {}


def records_to_original_(rec, pad_infos):
    """bboxes_to_original_ for a whole batch of records in place: pad_infos = one (ori w, ori h, tl x, tl y, imw, imh)
    per image (a row of ones-and-zeros (1, 1, 0, 0, 1, 1) leaves an image's boxes unchanged bit for bit)."""
    bbox = rec['bbox']
    require_gpu(bbox, 'records_to_original_')
    B, K = bbox.shape[0], bbox.shape[1]
    pad = torch.tensor([[float(v) for v in p] if p is not None else [1.0, 1.0, 0.0, 0.0, 1.0, 1.0] for p in pad_infos],
                       dtype=torch.float32).to(bbox.device, non_blocking=True)
    assert pad.shape == (B, 6) and bbox.stride(1) == 4 and bbox.stride(2) == 1
    code = _lib.lib().mydet_bboxes_to_original_batched_f32(_ptr(bbox), bbox.stride(0), _ptr(rec['count']), rec['count'].stride(0),
                                                           B, K, _ptr(pad), _stream())
    _lib.check(code, 'mydet_bboxes_to_original_batched_f32')
    return rec


def detections_to_json(bbox, score, cls, count=None, cat_table=None):
    """The numbers of ImageObjects.to_json for B x K detection slots in one launch: returns (rows float64 [B,K,5] =
    x1, y1, w, h, score in the reference's double arithmetic, category int64 [B,K]).  bbox [B,K,4] (any image
    stride), score [B,K], cls [B,K] int64, count [B] int32 or None, cat_table int64 [n] or None."""
    require_gpu(bbox, 'detections_to_json')
    B, K = bbox.shape[0], bbox.shape[1]
    out = torch.empty((B, K, 5), dtype=torch.float64, device=bbox.device)
    cat = torch.empty((B, K), dtype=torch.int64, device=bbox.device)
    if K == 0:
        return out, cat
    assert bbox.stride(1) == 4 and bbox.stride(2) == 1 and score.stride(1) == 1 and cls.stride(1) == 1
    code = _lib.lib().mydet_detections_to_json_f64(_ptr(bbox), bbox.stride(0), _ptr(score), score.stride(0), _ptr(cls),
                                                   cls.stride(0), _ptr(count), count.stride(0) if count is not None else 0,
                                                   B, K, _ptr(cat_table), cat_table.numel() if cat_table is not None else 0,
                                                   _ptr(out), _ptr(cat), _stream())
    _lib.check(code, 'mydet_detections_to_json_f64')
    return out, cat


IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def preprocess_u8(img_u8, out_hw, input_format):
    """uint8 [B,H,W,3] (or [H,W,3]) device tensor -> float32 [B,3,Hp,Wp]: zero-pad right/bottom, /255, and for
    'RGB_1_norm' the ImageNet normalisation -- the tensor the reference builds on the host with tvf.pad +
    tvf.to_tensor + format_tensor_img (api/detection.py:158-163)."""
    require_gpu(img_u8, 'preprocess_u8')
    if img_u8.dim() == 3:
        img_u8 = img_u8.unsqueeze(0)
    assert img_u8.dtype == torch.uint8 and img_u8.shape[-1] == 3
    img_u8 = img_u8.contiguous()
    B, H, W, _ = img_u8.shape
    Hp, Wp = out_hw
    if input_format not in ('RGB_1', 'RGB_1_norm'):
        raise NotImplementedError()
    norm = 1 if input_format == 'RGB_1_norm' else 0
    mean = np.asarray(IMAGENET_MEAN, dtype=np.float32)
    std = np.asarray(IMAGENET_STD, dtype=np.float32)
    out = torch.empty((B, 3, Hp, Wp), dtype=torch.float32, device=img_u8.device)
    code = _lib.lib().mydet_preprocess_u8_f32(_ptr(img_u8), B, H, W, _ptr(out), Hp, Wp, norm,
                                              ctypes.c_void_p(mean.ctypes.data), ctypes.c_void_p(std.ctypes.data), _stream())
    _lib.check(code, 'mydet_preprocess_u8_f32')
    return out
