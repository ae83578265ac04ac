"""GPU parity: each HIP kernel, called through the C ABI (mydetection_amd.ops -> ctypes), against
the CPU oracle / committed golden vectors on the same seeded inputs."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'these tests need the MI355X'
    from mydetection_amd import _lib
    _lib.lib()                                   # fail loudly if the HIP library is missing
    return torch.device('cuda:0')


def _conv_case(dev, B, Cin, Cout, k, s, H, W, act, residual=False, bias_only=False, pad=None, seed=0, wino=False, wino4=False):
    from mydetection_amd import ops
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    scale = None if bias_only else torch.rand(Cout, generator=g) + 0.5
    shift = torch.randn(Cout, generator=g) * 0.1
    pad = pad or ((k - 1) // 2,) * 4                      # (top, left, bottom, right)
    xp = F.pad(x, (pad[1], pad[3], pad[0], pad[2]))
    ref = F.conv2d(xp.double(), w.double(), None, s)
    ref = ref * (scale.double().view(1, -1, 1, 1) if scale is not None else 1.0) + shift.double().view(1, -1, 1, 1)
    if act == 1:
        ref = F.leaky_relu(ref, 0.1)
    elif act == 2:
        ref = ref * torch.sigmoid(ref)
    res = None
    if residual:
        res = torch.randn(ref.shape, generator=g)
        ref = ref + res.double()
    w_dev = w.permute(0, 2, 3, 1).contiguous().to(dev)
    u = None
    u4 = None
    if wino:
        u = ops.wino_weights(w_dev)
        assert u is not None and ops.WINOGRAD
    if wino4:
        u4 = ops.wino4_weights(w_dev)
        assert u4 is not None and ops.WINOGRAD and ops.WINOGRAD4
    y = ops.conv2d(x.to(dev).contiguous(memory_format=torch.channels_last), w_dev,
                   scale.to(dev) if scale is not None else None, shift.to(dev), k, s, pad, act,
                   residual=res.to(dev).contiguous(memory_format=torch.channels_last) if residual else None, wino=u, wino4=u4)
    assert tuple(y.shape) == tuple(ref.shape)
    err = (y.cpu().double() - ref).abs().max().item()
    # F(4x4,3x3) transform constants reach 8: ~14x the direct form's round-off (tests/test_winograd_algebra.py), still
    # 30x inside the 1e-4 the detections are held to
    tol = (6e-5 if wino4 else 2e-5) * max(1.0, ref.abs().max().item())
    assert err <= tol, f'conv mismatch {err} > {tol}'


@pytest.mark.parametrize('case', [
    dict(B=2, Cin=32, Cout=64, k=3, s=2, H=32, W=32, act=1),                   # stride-2 downsample
    dict(B=2, Cin=64, Cout=32, k=1, s=1, H=16, W=24, act=1),                   # DarkBlock 1x1, BN=32 tile
    dict(B=1, Cin=32, Cout=64, k=3, s=1, H=16, W=16, act=1, residual=True),    # DarkBlock 3x3 + residual
    dict(B=2, Cin=128, Cout=256, k=3, s=1, H=20, W=20, act=1, residual=True),  # 128x128 / 128x64 tiles
    dict(B=3, Cin=256, Cout=255, k=1, s=1, H=13, W=11, act=0, bias_only=True), # YOLO head: ragged M and N
    dict(B=1, Cin=768, Cout=256, k=1, s=1, H=8, W=8, act=1),                   # FPN concat input, small M
    dict(B=4, Cin=512, Cout=1024, k=3, s=1, H=10, W=10, act=1),                # deep K = 4608
    dict(B=2, Cin=24, Cout=144, k=1, s=1, H=12, W=12, act=2),                  # generic-K path (Cin % 32 != 0), swish
    dict(B=1, Cin=88, Cout=88, k=3, s=1, H=10, W=10, act=0, bias_only=True),   # generic-K 3x3
    dict(B=1, Cin=32, Cout=32, k=3, s=2, H=16, W=16, act=2, pad=(0, 0, 1, 1)), # static-SAME asymmetric pad
    dict(B=32, Cin=64, Cout=128, k=3, s=2, H=64, W=64, act=1),                 # big grid (XCD remap path)
])
def test_conv_igemm_vs_fp64(dev, case):
    _conv_case(dev, **case)


@pytest.mark.parametrize('case', [
    dict(B=32, Cin=32, Cout=64, k=3, s=2, H=64, W=64, act=1),                    # 128 x 64 tile (Cout <= 64), the first stride-2 layer's shape
    dict(B=32, Cin=128, Cout=256, k=3, s=2, H=40, W=40, act=1),                  # 128 x 128 tiles, 72 slabs
    dict(B=16, Cin=256, Cout=128, k=1, s=1, H=40, W=40, act=1, residual=True),   # 1x1 flat path + residual, 200 tiles
    dict(B=9, Cin=512, Cout=255, k=1, s=1, H=31, W=33, act=0, bias_only=True),   # head conv: ragged rows and channels (255), bias only
    dict(B=9, Cin=64, Cout=192, k=3, s=2, H=77, W=53, act=2),                    # odd sizes, swish, a half-empty second channel tile
    dict(B=40, Cin=1024, Cout=512, k=1, s=1, H=20, W=20, act=1),                 # 125 x 4 = 500 tiles: one round of 512
    dict(B=33, Cin=512, Cout=1024, k=3, s=2, H=40, W=40, act=1),                 # 288 slabs
    dict(B=33, Cin=512, Cout=1024, k=3, s=2, H=40, W=40, act=1, form='wide'),    # wide form (128 x 256 tiles, DMA-fed weights): 104 x 4 tiles, 288 slabs
    dict(B=69, Cin=128, Cout=256, k=3, s=2, H=64, W=64, act=1, residual=True, form='wide'),   # 552 tiles = two rounds of 256 + 40 cut along K (fixup launch)
    dict(B=12, Cin=256, Cout=320, k=1, s=1, H=27, W=29, act=2, form='wide'),     # ragged rows, a quarter-full second channel tile (320)
    dict(B=32, Cin=128, Cout=256, k=3, s=2, H=40, W=40, act=1, form='waves8'),   # 8-wave workgroups (wave tile 64 x 32)
    dict(B=9, Cin=512, Cout=255, k=1, s=1, H=31, W=33, act=0, bias_only=True, form='waves8'),
    dict(B=16, Cin=1152, Cout=192, k=1, s=1, H=20, W=20, act=0, residual=True, gate=True),     # MBConv project conv: SE gate on the A operand, small grid cut along K
    dict(B=16, Cin=672, Cout=112, k=1, s=1, H=40, W=40, act=0, gate=True),                     # ragged channel tile (112 of 128), gate, no skip
    dict(B=8, Cin=480, Cout=80, k=1, s=1, H=40, W=40, act=0, residual=True, gate=True),
    dict(B=5, Cin=96, Cout=64, k=1, s=1, H=33, W=31, act=0, gate=True),                        # 128 x 64 tile, ragged rows crossing image borders
    dict(B=16, Cin=256, Cout=128, k=1, s=1, H=40, W=40, act=1, residual=True, strided=True),      # input a channel slice of a wider map (ldx > Cin), output rows padded (ldy > Cout)
    dict(B=12, Cin=64, Cout=160, k=3, s=2, H=64, W=64, act=1, strided=True),
    dict(B=8, Cin=40, Cout=240, k=1, s=1, H=80, W=80, act=2),                                    # Cin % 16 == 8: the last slab of a 1x1 layer zero-filled (EfficientNet 40 -> 240)
    dict(B=4, Cin=24, Cout=144, k=1, s=1, H=50, W=46, act=2, strided=True),                      # Cin % 16 == 8 again, a channel slice: what lies behind the 24 channels must not enter
    dict(B=8, Cin=192, Cout=1152, k=1, s=1, H=10, W=10, act=2),                                # 63 tiles of 128 rows: the 64-row tile without a gate, swish, ragged last row tile
    dict(B=40, Cin=672, Cout=112, k=1, s=1, H=40, W=40, act=0, residual=True, gate=True),        # 500 tiles: the 128-row tile with the gate (the four above run 64-row tiles)
])
def test_conv_split_bf16_vs_fp64(dev, case, monkeypatch):
    """The implicit GEMM on the bfloat16 matrix instructions with float32-exact split operands (conv_igemm_b3_kernel;
    ops.conv2d(..., b3=)): held to the SAME 2e-5 * max|y| against float64 as the float32-instruction kernel, agrees with that
    kernel to float32 round-off, is bit-repeatable, and really ran (ops.b3_takes for the shape)."""
    from mydetection_amd import ops, _lib
    form = case.get('form')
    if form:                                               # the two opt-in forms of the launcher (read once per process otherwise)
        monkeypatch.setenv('MYDET_B3_WIDE' if form == 'wide' else 'MYDET_B3_WAVES', '1' if form == 'wide' else '8')
    assert _lib.lib().mydet_conv_b3_reload_tuning() == {None: 0, 'wide': 1, 'waves8': 2}[form]
    try:
        _split_bf16_case(dev, case)
    finally:
        monkeypatch.undo()
        assert _lib.lib().mydet_conv_b3_reload_tuning() == 0


@pytest.mark.parametrize('case', [
    dict(B=4, Cin=32, Cout=64, s=2, H=64, W=64, act=1),                         # BN = 64 form, two slabs, whole tiles
    dict(B=3, Cin=64, Cout=128, s=2, H=80, W=96, act=1),                        # BN = 128 form (single weight buffer)
    dict(B=2, Cin=128, Cout=256, s=2, H=40, W=40, act=1),                       # 20 x 20 outputs: ragged tiles in both directions, two channel tiles
    dict(B=3, Cin=48, Cout=192, s=2, H=37, W=53, act=0, bias_only=True),        # odd sizes, no activation, half-empty second channel tile
    dict(B=2, Cin=32, Cout=64, s=1, H=48, W=64, act=1, residual=True),          # stride 1 (the 32 -> 64 layer of the first DarkBlock) + residual
    dict(B=2, Cin=64, Cout=160, s=1, H=21, W=35, act=1, strided=True),          # stride 1, BN = 128, ragged, input a channel slice / padded output rows
    dict(B=5, Cin=16, Cout=40, s=2, H=30, W=18, act=1, residual=True, strided=True),   # one slab, ragged channels (40 of 64), residual
    dict(B=32, Cin=64, Cout=128, s=2, H=64, W=64, act=1),                       # 4 096 workgroups (XCD remap, several rounds)
    dict(B=3, Cin=64, Cout=128, s=2, H=80, W=80, act=1),                        # 40 x 40 outputs: two 8 x 16 columns + 16 x 8 STRIP tiles (the last one half empty)
    dict(B=3, Cin=32, Cout=192, s=2, H=40, W=40, act=1, residual=True),         # 20 x 20 outputs: one column + ONE 32 x 4 strip tile per image, ragged channels
    dict(B=2, Cin=48, Cout=64, s=2, H=47, W=48, act=0, strided=True),           # 24 x 24: 64-channel tiles with a 16 x 8 strip, odd height
    dict(B=2, Cin=16, Cout=40, s=2, H=70, W=8, act=1),                          # 4 columns only: no whole tile at all, 32 x 4 strips over 35 rows
])
def test_conv_p3_vs_fp64(dev, case):
    """conv_p3_kernel (csrc/conv_p3.hip: 3x3 conv with the workgroup's input patch resident in LDS, split-bf16 operands): held to the
    same 2e-5 * max|y| against float64 as conv_igemm_kernel / conv_igemm_b3_kernel, agrees with the float32 kernel to float32
    round-off, and is bit-repeatable."""
    from mydetection_amd import ops
    B, Cin, Cout, s, H, W = (case[n] for n in ('B', 'Cin', 'Cout', 's', 'H', 'W'))
    g = torch.Generator().manual_seed(3 * Cin + Cout + s)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    scale = None if case.get('bias_only') else torch.rand(Cout, generator=g) + 0.5
    shift = torch.randn(Cout, generator=g) * 0.1
    ref = F.conv2d(F.pad(x.double(), (1, 1, 1, 1)), w.double(), None, s)
    ref = ref * (scale.double().view(1, -1, 1, 1) if scale is not None else 1.0) + shift.double().view(1, -1, 1, 1)
    if case['act'] == 1:
        ref = F.leaky_relu(ref, 0.1)
    res = None
    if case.get('residual'):
        res = torch.randn(ref.shape, generator=g)
        ref = ref + res.double()
    xd = x.to(dev).contiguous(memory_format=torch.channels_last)
    if case.get('strided'):
        wide = torch.randn(B, Cin + 24, H, W, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
        wide[:, 8:8 + Cin] = xd
        xd = wide[:, 8:8 + Cin]
    wd = w.permute(0, 2, 3, 1).contiguous().to(dev)
    w3 = ops.split_bf16(wd)
    rd = res.to(dev).contiguous(memory_format=torch.channels_last) if res is not None else None
    kw = dict(out_ld=Cout + 12) if case.get('strided') else {}
    sc_d, sh_d = (scale.to(dev) if scale is not None else None), shift.to(dev)
    y = ops.conv3x3_p3(xd, w3, sc_d, sh_d, s, case['act'], residual=rd, **kw)
    assert y is not None and tuple(y.shape) == tuple(ref.shape)
    y2 = ops.conv3x3_p3(xd, w3, sc_d, sh_d, s, case['act'], residual=rd, **kw)
    assert torch.equal(y, y2)
    y32 = ops.conv2d(xd, wd, sc_d, sh_d, 3, s, (1, 1, 1, 1), case['act'], residual=rd, **kw)
    tol = 2e-5 * ref.abs().max().item()
    err = (y.cpu().double() - ref).abs().max().item()
    err32 = (y32.cpu().double() - ref).abs().max().item()
    assert err <= tol, f'{case}: {err:.2e} vs tol {tol:.2e} (float32 kernel: {err32:.2e})'
    assert err <= 4.0 * err32 + 1e-6, (err, err32)
    if case.get('strided'):              # nothing written behind the Cout channels of a padded output row
        assert ops.nhwc_ld(y) == Cout + 12


def test_split_bf16_non_finite_semantics(dev):
    """Deliberate deviation, pinned (include/mydet.h: mydet_conv2d_igemm_b3_f32): the split-bf16
    kernels are for finite tensors.  Where the float32 kernel and the reference propagate an infinite activation as inf, the
    three-piece split forms inf - inf: NaN; a finite |x| > 3.3962e38 rounds up to a bfloat16 inf and becomes NaN as well; NaN stays
    NaN; every output that does not touch the offending element is unaffected and finite."""
    from mydetection_amd import ops
    g = torch.Generator().manual_seed(9)
    B, Cin, Cout, H = 2, 64, 128, 16
    x = torch.randn(B, Cin, H, H, generator=g)
    w = (torch.randn(Cout, 1, 1, Cin, generator=g) / Cin ** 0.5).to(dev)
    sc, sh = torch.ones(Cout, device=dev), torch.zeros(Cout, device=dev)
    x[0, 3, 2, 5] = float('inf')
    x[0, 7, 9, 1] = 3.4e38                         # finite in float32, past the midpoint above the largest bfloat16 (3.3962e38): bf16 inf
    x[1, 0, 0, 0] = float('nan')
    xd = x.to(dev).contiguous(memory_format=torch.channels_last)
    w3 = ops.split_bf16(w)
    y3 = ops.conv2d(xd, w, sc, sh, 1, 1, (0, 0, 0, 0), ops.ACT_NONE, b3=w3, b3_min_rows=1)
    y32 = ops.conv2d(xd, w, sc, sh, 1, 1, (0, 0, 0, 0), ops.ACT_NONE)
    assert torch.isinf(y32[0, :, 2, 5]).all() and torch.isnan(y3[0, :, 2, 5]).all()          # inf -> NaN (float32 kernel: inf)
    assert torch.isfinite(y32[0, :, 9, 1]).all() and torch.isnan(y3[0, :, 9, 1]).all()        # 3.4e38 -> NaN (float32 kernel: finite)
    assert torch.isnan(y32[1, :, 0, 0]).all() and torch.isnan(y3[1, :, 0, 0]).all()           # NaN -> NaN in both
    clean = torch.ones(B, H, H, dtype=torch.bool, device=dev)
    clean[0, 2, 5] = clean[0, 9, 1] = clean[1, 0, 0] = False
    a, b = y3.permute(0, 2, 3, 1)[clean], y32.permute(0, 2, 3, 1)[clean]
    assert torch.isfinite(a).all() and (a - b).abs().max().item() <= 2e-5 * b.abs().max().item()


def _split_bf16_case(dev, case):
    from mydetection_amd import ops
    B, Cin, Cout, k, s, H, W = (case[n] for n in ('B', 'Cin', 'Cout', 'k', 's', 'H', 'W'))
    g = torch.Generator().manual_seed(Cin + Cout)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    scale = None if case.get('bias_only') else torch.rand(Cout, generator=g) + 0.5
    shift = torch.randn(Cout, generator=g) * 0.1
    p = (k - 1) // 2
    gate = torch.rand(B, Cin, generator=g) if case.get('gate') else None
    xg = x.double() * gate.double().view(B, Cin, 1, 1) if gate is not None else x.double()
    ref = F.conv2d(F.pad(xg, (p, p, p, p)), w.double(), None, s)
    ref = ref * (scale.double().view(1, -1, 1, 1) if scale is not None else 1.0) + shift.double().view(1, -1, 1, 1)
    if case['act'] == 1:
        ref = F.leaky_relu(ref, 0.1)
    elif case['act'] == 2:
        ref = ref * torch.sigmoid(ref)
    res = None
    if case.get('residual'):
        res = torch.randn(ref.shape, generator=g)
        ref = ref + res.double()
    Ho, Wo = ref.shape[2:]
    assert ops.b3_takes(B * Ho * Wo, Cin, Cout, k, min_rows=1, min_cout=64)        # (the kernel, not the dispatch rule, is under test)
    xd = x.to(dev).contiguous(memory_format=torch.channels_last)
    if case.get('strided'):             # the same values as channels [8, 8 + Cin) of a map with 24 more channels
        wide = torch.randn(B, Cin + 24, H, W, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
        wide[:, 8:8 + Cin] = xd
        xd = wide[:, 8:8 + Cin]
        assert ops.to_nhwc(xd)[1] == Cin + 24
    wd = w.permute(0, 2, 3, 1).contiguous().to(dev)
    w3 = ops.split_bf16(wd)
    # the operand's layout (slab-major, 256-row padding, DMA swizzle): undone here, p0 + p1 + p2 == w to 2^-27
    K, CoutP = wd.numel() // Cout, (Cout + 255) // 256 * 256
    Kp = (K + 15) // 16 * 16
    pl = w3.view(torch.bfloat16).float().view(Kp // 16, 3, CoutP // 32, 64, 8).sum(1)          # [kt][blk][unit][8]
    rr = torch.arange(32, device=dev)
    unit = torch.stack([2 * rr + (h ^ ((rr >> 2) & 1)) for h in (0, 1)], 1)                    # [row in block][k-half]
    rows = pl[:, :, unit]                                                                     # [kt][blk][32][2][8]
    full = rows.permute(1, 2, 0, 3, 4).reshape(CoutP, Kp)
    back = full[:Cout, :K].reshape(wd.shape)
    assert (back - wd).abs().max().item() <= 2.0 ** -24 * wd.abs().max().item()
    assert not bool(full[Cout:].any()) and not bool(full[:, K:].any())
    kw = dict(residual=res.to(dev).contiguous(memory_format=torch.channels_last) if res is not None else None)
    if gate is not None:
        kw['gate'] = gate.to(dev)
    if case.get('strided'):
        kw['out_ld'] = Cout + 12
    args = (xd, wd, scale.to(dev) if scale is not None else None, shift.to(dev), k, s, (p, p, p, p), case['act'])
    ops.TIMER = ops.KernelTimer()
    p3_was, ops.CONV_P3 = ops.CONV_P3, False            # (conv_igemm_b3_kernel is under test: the patch-resident 3x3 kernel has its own test)
    try:
        y3 = ops.conv2d(*args, b3=w3, b3_min_rows=1, **kw).clone()
    finally:
        timer, ops.TIMER = ops.TIMER, None
        ops.CONV_P3 = p3_was
    assert 'conv_igemm_b3' in timer.spans and 'conv_igemm' not in timer.spans
    y32 = ops.conv2d(*args, **kw)
    tol = 2e-5 * ref.abs().max().item()
    e3, e32 = (y3.cpu().double() - ref).abs().max().item(), (y32.cpu().double() - ref).abs().max().item()
    assert e3 <= tol, (e3, e32, tol)
    assert e3 <= 4.0 * e32 + 1e-6, f'split-bf16 error {e3:.2e} vs float32-instruction error {e32:.2e}'
    ops.CONV_P3 = False
    try:
        for _ in range(2):
            assert torch.equal(ops.conv2d(*args, b3=w3, b3_min_rows=1, **kw), y3)
    finally:
        ops.CONV_P3 = p3_was


@pytest.mark.parametrize('shape', [(2, 256, 512, 256, 20, 20),      # 28 x 4 tiles, K = 24 slabs: the small-grid K cut + CAT
                                   (1, 256, 512, 256, 10, 10),      # the same layer at batch 1 (7 x 4 tiles cut along K)
                                   (1, 128, 256, 128, 32, 32),      # batch 1, 12 slabs: uncut
                                   (3, 64, 32, 255, 6, 10),         # ragged rows and channels
                                   (32, 128, 256, 128, 40, 40)])    # the headline's P3 layer: big grid
def test_conv1x1_upcat_equals_two_launches(dev, shape):
    """cbl_0(cat((up2x(pre), x), 1)) of YOLOBranch in ONE launch (mydet_conv1x1_upcat_f32: the concatenation is read on the fly)
    == upsample_concat + conv2d bit for bit (same k order, same tile shape) -- big grids, the small-grid K-cut path at batch
    1 and 2, ragged rows and channels -- and within round-off of float64; shapes it does not cover return None, among them
    every shape the regular dispatch rule would not run on the 64 x 64 x 32 tile the fused launch exists for."""
    from mydetection_amd import ops
    B, C1, C2, Cout, Ha, Wa = shape
    g = torch.Generator().manual_seed(21)
    lo = torch.randn(B, C1, Ha, Wa, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    hi = torch.randn(B, C2, 2 * Ha, 2 * Wa, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Cout, 1, 1, C1 + C2, generator=g) / (C1 + C2) ** 0.5).to(dev)
    sc, sh = (torch.rand(Cout, generator=g) + 0.5).to(dev), torch.randn(Cout, generator=g).to(dev)
    y = ops.conv1x1_upcat(lo, hi, w, sc, sh, ops.ACT_LEAKY)
    cat = ops.upsample_concat(lo, (2 * Ha, 2 * Wa), hi)
    y2 = ops.conv2d(cat, w, sc, sh, 1, 1, (0, 0, 0, 0), ops.ACT_LEAKY)
    assert y is not None and torch.equal(y, y2)
    ref = torch.nn.functional.leaky_relu(
        torch.einsum('bchw,oc->bohw', torch.cat((torch.nn.functional.interpolate(lo.double().cpu(), scale_factor=2, mode='nearest'),
                                                 hi.double().cpu()), 1), w.double().cpu().view(Cout, -1))
        * sc.double().cpu().view(1, -1, 1, 1) + sh.double().cpu().view(1, -1, 1, 1), 0.1)
    assert (y.double().cpu() - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()
    assert ops.conv1x1_upcat(lo[:, :24], hi, w[..., :24 + C2].contiguous(), sc, sh, ops.ACT_LEAKY) is None      # C1 % 32 != 0
    assert ops.conv1x1_upcat(lo, hi, w, sc, sh, ops.ACT_SWISH) is None
    # <= 64 output channels: conv2d takes its 128 x 64 tile there, so the fused launch declines (the caller's two launches
    # then ARE the result; ConvBnLeaky(upcat_lo=...) does exactly that)
    assert ops.conv1x1_upcat(lo, hi, w[:64].contiguous(), sc[:64].contiguous(), sh[:64].contiguous(), ops.ACT_LEAKY) is None


@pytest.mark.parametrize('case', [
    dict(B=1, Cin=32, Cout=64, H=16, W=16, act=1, residual=True),      # DarkBlock 3x3 + residual, one workgroup row
    dict(B=2, Cin=128, Cout=256, H=20, W=20, act=1, residual=True),    # 4 channel blocks, tiles straddle images
    dict(B=4, Cin=512, Cout=1024, H=10, W=10, act=1),                  # deep K: 64 slabs
    dict(B=3, Cin=64, Cout=128, H=13, W=11, act=1),                    # odd H and W: half-empty edge tiles
    dict(B=1, Cin=88, Cout=88, H=10, W=10, act=0, bias_only=True),     # Cout % 64 != 0 (zero-padded U rows)
    dict(B=2, Cin=88, Cout=84, H=5, W=5, act=2),                       # ragged everything, swish
    dict(B=32, Cin=64, Cout=128, H=40, W=40, act=1, residual=True),    # big grid (XCD remap path)
    dict(B=32, Cin=128, Cout=256, H=40, W=40, act=1, residual=True),   # stream-K, 64-tile shape: 3.125 items/workgroup
    dict(B=16, Cin=64, Cout=128, H=80, W=80, act=1),                   # stream-K, 32-tile shape
    dict(B=24, Cin=136, Cout=200, H=37, W=37, act=2, residual=True),   # stream-K with ragged tiles, channels and K=17 slabs
    dict(B=1, Cin=128, Cout=128, H=8, W=8, act=1, residual=True),      # small grid cut along K: 2 items x 16 slabs on 32 workgroups
    dict(B=1, Cin=512, Cout=1024, H=16, W=16, act=1, residual=True),   # batch-1 deep layer: 16 items x 64 slabs over the whole chip
    dict(B=2, Cin=256, Cout=192, H=6, W=7, act=0, bias_only=True),     # small, ragged: 3 items x 32 slabs on 96 workgroups
])
def test_conv_winograd_vs_fp64(dev, case):
    """Fused Winograd F(2x2,3x3) kernel (3x3, stride 1, pad 1) against the same float64 reference and tolerance
    as the direct implicit-GEMM kernel."""
    _conv_case(dev, k=3, s=1, wino=True, **case)


@pytest.mark.parametrize('case', [
    dict(B=2, Cin=128, Cout=256, H=20, W=20, act=1, residual=True),    # 4 channel blocks, tiles straddle images
    dict(B=4, Cin=512, Cout=1024, H=12, W=12, act=1),                  # deep K: 128 slabs
    dict(B=3, Cin=64, Cout=128, H=13, W=11, act=1),                    # odd H and W: partial edge tiles
    dict(B=1, Cin=88, Cout=88, H=10, W=10, act=0, bias_only=True),     # Cout % 64 != 0 (zero-padded U rows), Cin % 8 != 0
    dict(B=2, Cin=132, Cout=84, H=5, W=7, act=2, residual=True),       # ragged everything, swish
    dict(B=32, Cin=128, Cout=256, H=40, W=40, act=1, residual=True),   # big grid (XCD remap path)
    dict(B=1, Cin=256, Cout=512, H=32, W=32, act=1, residual=True),    # batch-1 layer
    dict(B=17, Cin=64, Cout=512, H=32, W=32, act=1, residual=True),    # K-cut tail: 544 items = one round + 32 items cut 4 ways (nk / 4)
    dict(B=24, Cin=136, Cout=200, H=37, W=37, act=2, residual=True),   # K-cut tail, ragged: 525 items, a 21-item block cut 8 ways
])
def test_conv_winograd4_vs_fp64(dev, case):
    """Fused Winograd F(4x4,3x3) kernel (3x3, stride 1, pad 1) against the float64 reference."""
    _conv_case(dev, k=3, s=1, wino4=True, **case)


@pytest.mark.parametrize('case', [
    dict(B=4, Cin=512, Cout=1024, H=12, W=12, act=1),                  # deep K, one round of workgroups
    dict(B=32, Cin=128, Cout=256, H=40, W=40, act=1, residual=True),   # several rounds, residual
    dict(B=8, Cin=256, Cout=512, H=20, W=20, act=0, bias_only=True),
])
def test_conv_winograd4_repeatable(dev, case):
    """The F(4x4,3x3) pair of launches gives bit-identical results run after run (every LDS stage is fenced by a DMA
    wait + barrier, the workspace is written whole before it is read): a race would show as run-to-run differences."""
    from mydetection_amd import ops
    g = torch.Generator().manual_seed(5)
    B, Cin, Cout, H, W = (case[k] for k in ('B', 'Cin', 'Cout', 'H', 'W'))
    x = torch.randn(B, Cin, H, W, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Cout, 3, 3, Cin, generator=g) / (Cin * 9) ** 0.5).to(dev)
    scale = None if case.get('bias_only') else (torch.rand(Cout, generator=g) + 0.5).to(dev)
    shift = (torch.randn(Cout, generator=g) * 0.1).to(dev)
    res = torch.randn(B, Cout, H, W, generator=g).to(dev).contiguous(memory_format=torch.channels_last) if case.get('residual') else None
    u4 = ops.wino4_weights(w)
    outs = [ops.conv2d(x, w, scale, shift, 3, 1, (1, 1, 1, 1), case['act'], residual=res, wino4=u4).clone() for _ in range(6)]
    direct = ops.conv2d(x, w, scale, shift, 3, 1, (1, 1, 1, 1), case['act'], residual=res)
    for o in outs[1:]:
        assert torch.equal(o, outs[0])
    assert (outs[0] - direct).abs().max().item() <= 1e-4 * max(1.0, direct.abs().max().item())



@pytest.mark.parametrize('case', [
    dict(B=17, Cin=64, Cout=512, H=32, W=32, act=1, residual=True),
    dict(B=24, Cin=136, Cout=200, H=37, W=37, act=2, residual=True),
    dict(B=32, Cin=256, Cout=512, H=40, W=40, act=1, residual=True),   # the headline's 40^2 layers: 1536 items + 64 cut 8 ways
    dict(B=32, Cin=512, Cout=1024, H=20, W=20, act=1, residual=True),  # the headline's 20^2 layers: 512 + 256 (2 ways) + 32 (8 ways, ragged block row)
])
def test_conv_winograd4_kcut_tail(dev, case, monkeypatch):
    """F(4x4) launches whose item count is whole rounds plus a small remainder run the remainder as K pieces + a fixup launch
    (conv_wino4.hip, launch rule in mydet_conv2d_wino4_f32).  Same sums in another association: equal to the uncut form
    (MYDET_W4_TAIL=0) within float32 round-off, and bit-identical run after run."""
    from mydetection_amd import ops
    g = torch.Generator().manual_seed(9)
    B, Cin, Cout, H, W = (case[k] for k in ('B', 'Cin', 'Cout', 'H', 'W'))
    x = torch.randn(B, Cin, H, W, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Cout, 3, 3, Cin, generator=g) / (Cin * 9) ** 0.5).to(dev)
    scale, shift = (torch.rand(Cout, generator=g) + 0.5).to(dev), (torch.randn(Cout, generator=g) * 0.1).to(dev)
    res = torch.randn(B, Cout, H, W, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    u4 = ops.wino4_weights(w)

    def run():
        return ops.conv2d(x, w, scale, shift, 3, 1, (1, 1, 1, 1), case['act'], residual=res, wino4=u4).clone()
    cut = [run() for _ in range(4)]
    from mydetection_amd import _lib
    monkeypatch.setenv('MYDET_W4_TAIL', '0')
    _lib.lib().mydet_wino4_reload_tuning()                # (the knobs are read once per process otherwise)
    try:
        plain = run()
    finally:
        monkeypatch.delenv('MYDET_W4_TAIL')
        _lib.lib().mydet_wino4_reload_tuning()
    for o in cut[1:]:
        assert torch.equal(o, cut[0])
    assert not torch.equal(cut[0], plain), 'the tail rule did not trigger on a shape chosen to trigger it'
    assert (cut[0] - plain).abs().max().item() <= 2e-5 * max(1.0, plain.abs().max().item())


def test_conv_winograd_unsupported_shapes_stay_direct(dev):
    from mydetection_amd import ops
    assert ops.wino_weights(torch.zeros(64, 3, 3, 12, device=dev)) is None     # Cin % 8
    assert ops.wino_weights(torch.zeros(255, 3, 3, 64, device=dev)) is None    # Cout % 4
    assert ops.wino_weights(torch.zeros(64, 1, 1, 64, device=dev)) is None     # not 3x3


def test_conv_igemm_output_slice_and_padded_ld(dev):
    """ldy > Cout (head padding) must leave the pad lanes untouched and slices view correctly."""
    from mydetection_amd import ops
    x = torch.randn(1, 64, 8, 8).to(dev).contiguous(memory_format=torch.channels_last)
    w = torch.randn(30, 64, 1, 1)
    y = ops.conv2d(x, w.permute(0, 2, 3, 1).contiguous().to(dev), None, None, 1, 1, (0, 0, 0, 0), 0)
    assert y.shape == (1, 30, 8, 8) and ops.nhwc_ld(y) == 32
    ref = F.conv2d(x.cpu().double(), w.double())
    assert (y.cpu().double() - ref).abs().max() < 1e-4


def test_conv_stem(dev):
    from mydetection_amd import ops
    g = torch.Generator().manual_seed(1)
    for layout in ('nchw', 'nhwc'):
        for s, pad in ((1, (1, 1, 1, 1)), (2, (0, 0, 1, 1))):
            x = torch.rand(2, 3, 34, 38, generator=g)
            w = torch.randn(32, 3, 3, 3, generator=g) * 0.2
            scale, shift = torch.rand(32, generator=g) + 0.5, torch.randn(32, generator=g) * 0.1
            xp = F.pad(x, (pad[1], pad[3], pad[0], pad[2]))
            ref = F.conv2d(xp.double(), w.double(), None, s) * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)
            ref = F.leaky_relu(ref, 0.1)
            xd = x.to(dev)
            if layout == 'nhwc':
                xd = xd.contiguous(memory_format=torch.channels_last)
            y = ops.conv2d_stem(xd, w.permute(0, 2, 3, 1).contiguous().to(dev), scale.to(dev), shift.to(dev), s, pad, 1)
            assert (y.cpu().double() - ref).abs().max() < 1e-5


def test_upsample_concat(dev):
    from mydetection_amd import ops
    a = torch.randn(2, 8, 5, 7)
    b = torch.randn(2, 12, 10, 14)
    y = ops.upsample_concat(a.to(dev), (10, 14), b.to(dev))
    ref = torch.cat((F.interpolate(a, size=(10, 14), mode='nearest'), b), dim=1)
    assert torch.equal(y.cpu(), ref)
    y2 = ops.upsample_concat(a.to(dev), (9, 13))                  # non-integer scale, no concat
    assert torch.equal(y2.cpu(), F.interpolate(a, size=(9, 13), mode='nearest'))


def test_ultralytics_blocks(dev):
    """Focus' space-to-depth (NCHW and channels-last images), SPP's pooled concatenation (bit-exact vs ATen: pure data
    movement / max), the 12-channel 3x3 Focus conv, and BottleneckCSP's split BatchNorm written into the halves of the
    concatenated buffer -- each module against the reference's formulation in float64 (external/ultralytics/common.py)."""
    from mydetection_amd import ops
    from mydetection_amd.external.ultralytics.common import BottleneckCSP, Focus, SPP
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 3, 12, 20, generator=g)
    ref = torch.cat([x[..., ::2, ::2], x[..., 1::2, ::2], x[..., ::2, 1::2], x[..., 1::2, 1::2]], 1)
    assert torch.equal(ops.space_to_depth(x.to(dev)).cpu(), ref)
    xl = x.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)                    # channels-last storage, same logical tensor
    assert torch.equal(ops.space_to_depth(xl.to(dev)).cpu(), ref)
    for hw in ((8, 8), (20, 20), (5, 13)):
        t = torch.randn(2, 16, *hw, generator=g)
        ref = torch.cat([t] + [F.max_pool2d(t, k, 1, k // 2) for k in (5, 9, 13)], 1)
        assert torch.equal(ops.spp_concat(t.to(dev)).cpu(), ref), hw

    def randomise(mod):
        with torch.no_grad():
            for n, p_ in list(mod.named_parameters()) + list(mod.named_buffers()):
                if n.endswith('running_var'):
                    p_.copy_(torch.rand(p_.shape, generator=g) + 0.5)
                elif n.endswith('num_batches_tracked'):
                    continue
                elif p_.dim() == 4:
                    p_.copy_(torch.randn(p_.shape, generator=g) / (p_.shape[1] * p_.shape[2] * p_.shape[3]) ** 0.5)
                else:
                    p_.copy_(torch.randn(p_.shape, generator=g) * 0.3 + (1.0 if n.endswith('bn.weight') else 0.0))
        return mod.eval()

    def bn(z, m):
        return F.batch_norm(z, m.running_mean.double(), m.running_var.double(), m.weight.double(), m.bias.double(), False, 0.0, m.eps)

    def conv_ref(z, m):             # common.Conv in float64
        return F.leaky_relu(bn(F.conv2d(z, m.conv.weight.double(), None, m.s, m.k // 2), m.bn), 0.1)

    foc = randomise(Focus(3, 48, k=3))
    xi = torch.rand(2, 3, 24, 40, generator=g)
    xd = xi.double()
    ref = conv_ref(torch.cat([xd[..., ::2, ::2], xd[..., 1::2, ::2], xd[..., ::2, 1::2], xd[..., 1::2, 1::2]], 1), foc.conv)
    y = foc.to(dev)(xi.to(dev)).cpu().double()
    assert (y - ref).abs().max() <= 2e-5 * ref.abs().max()

    for shortcut in (True, False):
        csp = randomise(BottleneckCSP(24, 24, n=2, shortcut=shortcut))
        t = torch.randn(2, 24, 9, 11, generator=g)
        td = t.double()
        h = conv_ref(td, csp.cv1)
        for b_ in csp.m:
            r = conv_ref(conv_ref(h, b_.cv1), b_.cv2)
            h = h + r if b_.add else r
        cat = torch.cat((F.conv2d(h, csp.cv3.weight.double()), F.conv2d(td, csp.cv2.weight.double())), 1)
        ref = conv_ref(F.leaky_relu(bn(cat, csp.bn), 0.1), csp.cv4)
        y = csp.to(dev)(t.to(dev)).cpu().double()
        assert (y - ref).abs().max() <= 3e-5 * ref.abs().max(), shortcut

    spp = randomise(SPP(32, 32))
    t = torch.randn(2, 32, 10, 10, generator=g)
    h = conv_ref(t.double(), spp.cv1)
    ref = conv_ref(torch.cat([h] + [F.max_pool2d(h, k, 1, k // 2) for k in (5, 9, 13)], 1), spp.cv2)
    y = spp.to(dev)(t.to(dev)).cpu().double()
    assert (y - ref).abs().max() <= 3e-5 * ref.abs().max()


def test_bboxes_iou_golden(dev, golden):
    from mydetection_amd.utils.bbox_ops import bboxes_iou
    g = golden('bbox_ops')
    out = bboxes_iou(torch.from_numpy(g['a']).to(dev), torch.from_numpy(g['b']).to(dev), xyxy=False)
    np.testing.assert_array_equal(out.cpu().numpy(), g['iou_cxcywh'])
    out = bboxes_iou(torch.from_numpy(g['a_xyxy']).to(dev), torch.from_numpy(g['b_xyxy']).to(dev), xyxy=True)
    np.testing.assert_array_equal(out.cpu().numpy(), g['iou_xyxy'])


def test_cxcywh_to_x1y1x2y2_golden(dev, golden):
    """The to-corners kernel against the reference's own cxcywh_to_x1y1x2y2 outputs (utils/bbox_ops.py:309-316), bit for
    bit; leading dimensions, a fifth column (carried over), the input untouched, an empty set."""
    from mydetection_amd.utils.bbox_ops import cxcywh_to_x1y1x2y2
    g = golden('bbox_ops')
    for src, want in ((g['a'], g['a_xyxy']), (g['b'], g['b_xyxy'])):
        t = torch.from_numpy(src).to(dev)
        out = cxcywh_to_x1y1x2y2(t)
        np.testing.assert_array_equal(out.cpu().numpy(), want)
        np.testing.assert_array_equal(t.cpu().numpy(), src)                 # a new tensor, as the reference's clone()
    both = torch.from_numpy(np.stack([g['a'][:30], g['b'][:30]])).to(dev)     # [2,30,4]
    np.testing.assert_array_equal(cxcywh_to_x1y1x2y2(both).cpu().numpy(), np.stack([g['a_xyxy'][:30], g['b_xyxy'][:30]]))
    five = np.concatenate([g['a'], np.arange(37, dtype=np.float32)[:, None]], 1)
    out5 = cxcywh_to_x1y1x2y2(torch.from_numpy(five).to(dev)).cpu().numpy()
    np.testing.assert_array_equal(out5[:, :4], g['a_xyxy'])
    np.testing.assert_array_equal(out5[:, 4], five[:, 4])
    assert tuple(cxcywh_to_x1y1x2y2(torch.empty((0, 4), device=dev)).shape) == (0, 4)
    # device and dtype of the input are kept, as in the reference: a CPU tensor comes back on the CPU, float64 / int64
    # boxes come back in their own dtype (computed in float32 on the GPU)
    cpu_out = cxcywh_to_x1y1x2y2(torch.from_numpy(g['a']))
    assert cpu_out.device.type == 'cpu' and cpu_out.dtype == torch.float32
    np.testing.assert_array_equal(cpu_out.numpy(), g['a_xyxy'])
    d64 = cxcywh_to_x1y1x2y2(torch.from_numpy(g['a']).double().to(dev))
    assert d64.dtype == torch.float64 and d64.is_cuda
    np.testing.assert_array_equal(d64.cpu().numpy(), g['a_xyxy'].astype(np.float64))
    ints = torch.tensor([[10, 20, 4, 8]], dtype=torch.int64)
    assert cxcywh_to_x1y1x2y2(ints).tolist() == [[8, 16, 12, 24]] and cxcywh_to_x1y1x2y2(ints).dtype == torch.int64


def _yolo_cfg():
    from mydetection_amd.models.general import load_config
    cfg = load_config('yolov3_80')
    cfg['model.fpn.out_strides'] = (8, 16, 32)
    cfg['model.fpn.out_channels'] = (256, 512, 1024)
    return cfg


def test_yolo_decode_golden(dev, golden):
    """Reference det-layer outputs (imported reference, tests/golden/detlayers.npz); class ids exact,
    boxes/scores to float32 round-off of exp/sigmoid."""
    from mydetection_amd.models.registry import get_det_layer
    g = golden('detlayers')
    cfg = _yolo_cfg()
    for lvl in (0, 1, 2):
        conv = torch.from_numpy(g[f'yolo_{lvl}_in']).to(dev)
        v = conv.view(conv.shape[0], 3, 85, *conv.shape[2:])
        raw = {'bbox': v[:, :, 0:4].permute(0, 1, 3, 4, 2), 'conf': v[:, :, 4:5].permute(0, 1, 3, 4, 2),
               'class': v[:, :, 5:].permute(0, 1, 3, 4, 2)}
        layer = get_det_layer(cfg)(level_i=lvl, cfg=cfg)
        p, loss = layer(raw, tuple(int(t) for t in g[f'yolo_{lvl}_img']), None)
        assert loss is None
        np.testing.assert_array_equal(p['class_idx'].cpu().numpy(), g[f'yolo_{lvl}_class_idx'])
        np.testing.assert_allclose(p['score'].cpu().numpy(), g[f'yolo_{lvl}_score'], rtol=2e-6, atol=1e-9)
        np.testing.assert_allclose(p['bbox'].cpu().numpy(), g[f'yolo_{lvl}_bbox'], rtol=2e-6, atol=1e-6)


def test_retina_fcos_decode_golden(dev, golden):
    from mydetection_amd import ops
    from mydetection_amd.models.detlayers._common import alloc_outputs, pack_pixel_major
    g = golden('detlayers')
    strides = [8, 16, 32, 64, 128]
    for lvl in (0, 3):
        bb_in = torch.from_numpy(g[f'retina_{lvl}_bbox_in']).to(dev)
        cl_in = torch.from_numpy(g[f'retina_{lvl}_class_in']).to(dev)
        B, A, H, W, _ = bb_in.shape
        box, ldb, _ = pack_pixel_major([bb_in], A)
        cls, ldc, _ = pack_pixel_major([cl_in], A)
        out = alloc_outputs(B, A * H * W, dev)
        ops.decode(ops.DECODE_RETINA, box, ldb, 4, 0, cls, ldc, 80, 0, 0, g[f'retina_{lvl}_anchor_wh'], A, 80, B, H, W,
                   strides[lvl], tuple(int(t) for t in g[f'retina_{lvl}_img']), *out, 0)
        np.testing.assert_array_equal(out[1].cpu().numpy(), g[f'retina_{lvl}_class_idx'])
        np.testing.assert_allclose(out[2].cpu().numpy(), g[f'retina_{lvl}_score'], rtol=2e-6, atol=1e-9)
        np.testing.assert_allclose(out[0].cpu().numpy(), g[f'retina_{lvl}_bbox'], rtol=2e-6, atol=1e-6)
    for lvl in (0, 2):
        bb_in = torch.from_numpy(g[f'fcos_{lvl}_bbox_in']).to(dev)
        cf_in = torch.from_numpy(g[f'fcos_{lvl}_conf_in']).to(dev)
        cl_in = torch.from_numpy(g[f'fcos_{lvl}_class_in']).to(dev)
        B, H, W, _ = bb_in.shape
        box, ldb, _ = pack_pixel_major([bb_in], 1)
        cls, ldc, per = pack_pixel_major([cf_in, cl_in], 1)          # conf at channel 0, classes 1..80
        out = alloc_outputs(B, H * W, dev)
        ops.decode(ops.DECODE_FCOS, box, ldb, 4, 0, cls, ldc, per, 1, 0, None, 1, 80, B, H, W, strides[lvl],
                   tuple(int(t) for t in g[f'fcos_{lvl}_img']), *out, 0)
        np.testing.assert_array_equal(out[1].cpu().numpy(), g[f'fcos_{lvl}_class_idx'])
        np.testing.assert_allclose(out[2].cpu().numpy(), g[f'fcos_{lvl}_score'], rtol=2e-6, atol=1e-9)
        np.testing.assert_allclose(out[0].cpu().numpy(), g[f'fcos_{lvl}_bbox'], rtol=2e-6, atol=1e-5)


def test_post_process_golden_bit_exact(dev, golden):
    """Filter/top-k/NMS on the reference's exact candidates: kept boxes, classes, scores and their
    order must be identical (integer/index work: bit-exact)."""
    from mydetection_amd.utils.structures import ImageObjects
    from oracle import postprocess as pp
    g = golden('postprocess')
    for name in g['names']:
        b, c, s = g[f'{name}_in_bboxes'], g[f'{name}_in_cats'], g[f'{name}_in_scores']
        conf, nms = float(g[f'{name}_conf']), float(g[f'{name}_nms'])
        d = ImageObjects(torch.from_numpy(b).to(dev), torch.from_numpy(c).to(dev), None, torch.from_numpy(s).to(dev),
                         'cxcywh', (512, 512))
        r = d.post_process(conf, nms)
        np.testing.assert_array_equal(r.cats.cpu().numpy(), g[f'{name}_cats'], err_msg=name)
        np.testing.assert_array_equal(r.scores.cpu().numpy(), g[f'{name}_scores'], err_msg=name)
        np.testing.assert_array_equal(r.bboxes.cpu().numpy(), g[f'{name}_bboxes'], err_msg=name)
        # candidate indices vs the oracle
        from mydetection_amd import ops
        if len(s):
            rec = ops.postprocess(d.bboxes[None], d.cats[None], d.scores[None], conf, nms)
            k = int(rec['count'][0])
            _, _, _, src = pp.post_process(b, c, s, conf, nms)
            np.testing.assert_array_equal(rec['index'][0, :k].cpu().numpy().astype(np.int64), src, err_msg=name)
            assert int(rec['index'][0, k:].abs().sum()) == 0


def test_post_process_batched_random_vs_oracle(dev):
    from mydetection_amd import ops
    from oracle import postprocess as pp
    rng = np.random.Generator(np.random.PCG64(11))
    B, N = 6, 9000
    b = np.empty((B, N, 4), np.float32)
    b[..., :2] = rng.random((B, N, 2), dtype=np.float32) * 200
    b[..., 2:] = rng.random((B, N, 2), dtype=np.float32) * 80 + 1
    c = rng.integers(0, 7, size=(B, N)).astype(np.int64)
    s = rng.random((B, N), dtype=np.float32)
    s[1] *= 0.01                                   # image 1: almost nothing passes
    s[2, 5:] = 0                                   # image 2: 5 candidates
    s[3, 100:300] = s[3, 50]                       # image 3: 200-way score tie at the top-k boundary region
    rec = ops.postprocess(torch.from_numpy(b).to(dev), torch.from_numpy(c).to(dev), torch.from_numpy(s).to(dev), 0.3, 0.4)
    for i in range(B):
        ob, oc, os_, src = pp.post_process(b[i], c[i], s[i], 0.3, 0.4)
        k = int(rec['count'][i])
        assert k == len(src)
        np.testing.assert_array_equal(rec['index'][i, :k].cpu().numpy().astype(np.int64), src)
        np.testing.assert_array_equal(rec['class_idx'][i, :k].cpu().numpy(), oc)
        np.testing.assert_array_equal(rec['score'][i, :k].cpu().numpy(), os_)
        np.testing.assert_array_equal(rec['bbox'][i, :k].cpu().numpy(), ob)


def test_post_process_suppression_chains_vs_oracle(dev):
    """Greedy selection by fixed-point rounds: chains where box i only overlaps box i+1 need one round per link, so
    a 400-box chain exhausts the round budget and takes the sequential form; short chains and a chain broken into
    classes settle early.  All must equal the oracle's sequential greedy NMS."""
    from mydetection_amd import ops
    from oracle import postprocess as pp
    n = 400
    b = np.zeros((4, n, 4), np.float32)
    b[..., 0] = 100 + 12.0 * np.arange(n, dtype=np.float32)      # 20-wide boxes shifted by 12: IoU(i,i+1)=0.25, (i,i+2)=0
    b[..., 1], b[..., 2], b[..., 3] = 50, 20, 20
    b[1, :, 0] = 100 + 6.0 * np.arange(n, dtype=np.float32)      # shift 6: IoU(i,i+1)=0.54, (i,i+2)=0.25, (i,i+3)=0.05
    s = np.tile(np.linspace(0.99, 0.5, n, dtype=np.float32), (4, 1))
    c = np.zeros((4, n), np.int64)
    c[2] = np.arange(n) // 7                                     # image 2: the chain of image 0 cut into 58 classes
    b[3] = b[1]
    c[3] = np.arange(n) % 3                                      # image 3: three interleaved chains
    for thr in (0.2, 0.5):
        rec = ops.postprocess(torch.from_numpy(b).to(dev), torch.from_numpy(c).to(dev), torch.from_numpy(s).to(dev), 0.3, thr)
        for i in range(4):
            ob, oc, os_, src = pp.post_process(b[i], c[i], s[i], 0.3, thr)
            k = int(rec['count'][i])
            assert k == len(src), (thr, i, k, len(src))
            np.testing.assert_array_equal(rec['index'][i, :k].cpu().numpy().astype(np.int64), src)
    # sanity of the construction: image 0 at thr 0.2 keeps every other box (a 400-link dependency chain)
    assert int(ops.postprocess(torch.from_numpy(b[:1]).to(dev), torch.from_numpy(c[:1]).to(dev),
                               torch.from_numpy(s[:1]).to(dev), 0.3, 0.2)['count'][0]) == n // 2


def test_nms_standalone_and_to_original(dev, golden):
    from mydetection_amd.utils.structures import ImageObjects
    from oracle import postprocess as pp
    g = golden('postprocess')
    b, c, s = (g[f'three_class_dense_in_{k}'] for k in ('bboxes', 'cats', 'scores'))
    d = ImageObjects(torch.from_numpy(b).to(dev), torch.from_numpy(c).to(dev), None, torch.from_numpy(s).to(dev))
    r = d.nms(0.3)
    keep = pp.class_aware_nms(b, c, s, 0.3)
    np.testing.assert_array_equal(r.scores.cpu().numpy(), s[keep])
    np.testing.assert_array_equal(r.cats.cpu().numpy(), c[keep])
    d2 = ImageObjects(torch.from_numpy(b.copy()).to(dev), torch.from_numpy(c).to(dev), None, torch.from_numpy(s).to(dev),
                      'cxcywh', (64, 64))
    pad_info = tuple(int(v) for v in g['to_original_pad_info'])
    d2.bboxes_to_original_(pad_info)
    np.testing.assert_array_equal(d2.bboxes.cpu().numpy(), g['to_original_bboxes'])
    assert d2.img_hw == (pad_info[1], pad_info[0])


# ------------------------------------------------------------------ EfficientNet / BiFPN family kernels
@pytest.mark.parametrize('k,s,pad,C,H,W,act', [(3, 1, (1, 1, 1, 1), 32, 20, 24, 2), (3, 2, (0, 0, 1, 1), 96, 16, 16, 2),
                                              (5, 1, (2, 2, 2, 2), 144, 12, 10, 2), (5, 2, (1, 1, 2, 2), 240, 10, 10, 2),
                                              (3, 1, (1, 1, 1, 1), 88, 5, 5, 0), (3, 1, (1, 1, 1, 1), 16, 41, 70, 2),
                                              (5, 1, (2, 2, 2, 2), 24, 19, 33, 2), (3, 1, (1, 1, 1, 1), 672, 40, 40, 2)])
def test_dwconv(dev, k, s, pad, C, H, W, act):
    from mydetection_amd import ops
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, C, H, W, generator=g)
    w = torch.randn(C, 1, k, k, generator=g) * 0.3
    bn = act != 0
    scale = torch.rand(C, generator=g) + 0.5 if bn else None
    shift = torch.randn(C, generator=g) * 0.1 if bn else None
    ref = F.conv2d(F.pad(x, (pad[1], pad[3], pad[0], pad[2])).double(), w.double(), None, s, 0, 1, C)
    if bn:
        ref = ref * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)
    if act == 2:
        ref = ref * torch.sigmoid(ref)
    y = ops.dwconv(x.to(dev), w.permute(2, 3, 0, 1).reshape(k, k, C).contiguous().to(dev),
                   scale.to(dev) if bn else None, shift.to(dev) if bn else None, k, s, pad, act)
    assert y.shape == ref.shape
    assert (y.cpu().double() - ref).abs().max() < 2e-5
    # squeeze-fused form: same y bit for bit, plus per-slice channel sums
    y2, partial = ops.dwconv(x.to(dev), w.permute(2, 3, 0, 1).reshape(k, k, C).contiguous().to(dev),
                             scale.to(dev) if bn else None, shift.to(dev) if bn else None, k, s, pad, act, squeeze=True)
    assert torch.equal(y2.contiguous(), y.contiguous())
    np.testing.assert_allclose(partial[:, :-1].sum(dim=1).cpu().double().numpy(), ref.sum(dim=(2, 3)).numpy(), rtol=1e-5, atol=1e-4)


def test_se_gate_and_gated_conv(dev):
    from mydetection_amd import ops
    g = torch.Generator().manual_seed(4)
    # (the two 1152-channel cases are small grids with long K: cut along K over the chip, gate and all)
    for C, Cse, H, W, Cout in ((96, 4, 40, 40, 24), (1152, 48, 5, 5, 192), (16, 4, 33, 17, 16), (1152, 48, 20, 20, 192)):
        x = torch.randn(3, C, H, W, generator=g)
        w1, b1 = torch.randn(Cse, C, generator=g) * 0.1, torch.randn(Cse, generator=g) * 0.1
        w2, b2 = torch.randn(C, Cse, generator=g) * 0.3, torch.randn(C, generator=g) * 0.1
        m = x.double().mean(dim=(2, 3))
        h = m @ w1.double().t() + b1.double()
        h = h * torch.sigmoid(h)
        gate_ref = torch.sigmoid(h @ w2.double().t() + b2.double())
        xd = x.to(dev).contiguous(memory_format=torch.channels_last)
        gate = ops.se_gate(ops.channel_sums(xd), H * W, w1.to(dev), b1.to(dev), w2.t().contiguous().to(dev), b2.to(dev))
        assert (gate.cpu().double() - gate_ref).abs().max() < 1e-5
        # project conv with the gate applied while staging A, + residual when shapes allow
        wp = torch.randn(Cout, C, 1, 1, generator=g) / C ** 0.5
        scale, shift = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.1
        res = torch.randn(3, Cout, H, W, generator=g)
        ref = F.conv2d(x.double() * gate_ref.view(3, C, 1, 1), wp.double()) * scale.double().view(1, -1, 1, 1) \
            + shift.double().view(1, -1, 1, 1) + res.double()
        y = ops.conv2d(xd, wp.permute(0, 2, 3, 1).contiguous().to(dev), scale.to(dev), shift.to(dev), 1, 1, (0, 0, 0, 0), 0,
                       residual=res.to(dev).contiguous(memory_format=torch.channels_last), gate=gate)
        assert (y.cpu().double() - ref).abs().max() < 3e-5 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize('B,H,W,chlast', [(2, 64, 64, False), (3, 100, 76, False), (1, 61, 95, True), (2, 640, 640, False)])
def test_stem_dw_fused(dev, B, H, W, chlast):
    """EfficientNet stem + block 0's depthwise conv as one launch (csrc/stem_dw.hip; the 32-channel stem map stays in LDS):
    against float64 (static SAME padding of a stride-2 3x3 conv on even and odd sizes, ragged tiles, both image
    layouts), against the two-launch path, and its per-tile channel sums against the map it wrote."""
    from mydetection_amd import ops
    g = torch.Generator().manual_seed(B * 1000 + H)
    x = torch.rand(B, 3, H, W, generator=g) * 2 - 1
    ws = torch.randn(32, 3, 3, 3, generator=g) * 0.3
    wd = torch.randn(32, 1, 3, 3, generator=g) * 0.3
    sc0, sh0 = torch.rand(32, generator=g) + 0.5, torch.randn(32, generator=g) * 0.2
    sc1, sh1 = torch.rand(32, generator=g) + 0.5, torch.randn(32, generator=g) * 0.2
    pt, pl = (1, 1) if H % 2 else (0, 0), (1, 1) if W % 2 else (0, 0)             # TF "SAME" for k3 s2: total 1 (even) / 2 (odd)
    pad = (pt[0] if H % 2 else 0, pl[0] if W % 2 else 0, 1, 1)
    xp = F.pad(x.double(), (pad[1], pad[3], pad[0], pad[2]))
    t = F.conv2d(xp, ws.double(), stride=2) * sc0.double().view(1, -1, 1, 1) + sh0.double().view(1, -1, 1, 1)
    t = t * torch.sigmoid(t)
    ref = F.conv2d(t, wd.double(), padding=1, groups=32) * sc1.double().view(1, -1, 1, 1) + sh1.double().view(1, -1, 1, 1)
    ref = ref * torch.sigmoid(ref)
    xd = x.to(dev).contiguous(memory_format=torch.channels_last) if chlast else x.to(dev)
    ws_o = ws.permute(0, 2, 3, 1).contiguous().to(dev)                            # OHWI
    wd_k = wd.permute(2, 3, 0, 1).reshape(3, 3, 32).contiguous().to(dev)          # [k,k,C]
    y, partial = ops.stem_dw(xd, ops.fold_scale(ws_o, sc0.to(dev)), sh0.to(dev), ops.fold_scale(wd_k, sc1.to(dev)), sh1.to(dev), pad)
    assert y.shape == ref.shape
    tol = 3e-5 * max(1.0, ref.abs().max().item())
    assert (y.cpu().double() - ref).abs().max() < tol
    np.testing.assert_allclose(partial[:, :-1].sum(dim=1).cpu().double().numpy(), y.cpu().double().sum(dim=(2, 3)).numpy(), rtol=1e-5, atol=1e-3)
    s = ops.conv2d_stem(xd, ws_o, sc0.to(dev), sh0.to(dev), 2, pad, ops.ACT_SWISH)
    y2 = ops.dwconv(s, wd_k, sc1.to(dev), sh1.to(dev), 3, 1, (1, 1, 1, 1), ops.ACT_SWISH)
    assert (y2 - y).abs().max().item() < tol
    y3, _ = ops.stem_dw(xd, ops.fold_scale(ws_o, sc0.to(dev)), sh0.to(dev), ops.fold_scale(wd_k, sc1.to(dev)), sh1.to(dev), pad)
    assert torch.equal(y3, y)


@pytest.mark.parametrize('C,Cse,S', [(96, 4, 800), (32, 8, 800), (144, 6, 200), (1920, 80, 6), (1152, 48, 15), (672, 28, 15),
                                      (240, 10, 50), (16, 4, 128), (480, 20, 1), (4100, 80, 3)])
def test_se_tail_from_slice_sums(dev, C, Cse, S):
    """The one-launch squeeze-excite tail (mean over S slice sums -> reduce conv + swish -> expand conv + sigmoid) on the
    slice counts / channel widths the EfficientNet blocks produce (S up to 800 tiles, C up to 1920; C = 4100 takes the
    two-launch fallback), against float64; twice: bit-identical (fixed summation order); the mean lands in slice S."""
    from mydetection_amd import ops
    g = torch.Generator().manual_seed(C + S)
    B, HW = 3, 16 * S
    partial = torch.randn(B, S + 1, C, generator=g) * 4
    w1, b1 = torch.randn(Cse, C, generator=g) / C ** 0.5, torch.randn(Cse, generator=g) * 0.1
    w2, b2 = torch.randn(C, Cse, generator=g) * 0.3, torch.randn(C, generator=g) * 0.1
    m = partial[:, :S].double().sum(dim=1) / HW
    h = m @ w1.double().t() + b1.double()
    h = h * torch.sigmoid(h)
    ref = torch.sigmoid(h @ w2.double().t() + b2.double())
    args = (HW, w1.to(dev), b1.to(dev), w2.t().contiguous().to(dev), b2.to(dev))
    pd = partial.to(dev)
    gate = ops.se_gate(pd, *args)
    assert (gate.cpu().double() - ref).abs().max() < 2e-6
    assert (pd[:, S].cpu().double() - m).abs().max() < 1e-5 * max(1.0, m.abs().max().item())
    assert torch.equal(ops.se_gate(partial.to(dev), *args), gate)


@pytest.mark.parametrize('kind,C,Cse,k,s,H,W', [
    ('dw', 1152, 48, 5, 1, 20, 20),        # LDS-tiled kernel, 36 channel chunks x 6 tiles per image (blocks 17-20)
    ('dw', 1920, 80, 3, 1, 20, 20),        # the widest block: Cse > 64 (two passes of the share computation)
    ('dw', 144, 6, 3, 1, 37, 41),          # ragged tiles, a last chunk of 16 channels
    ('dw', 16, 4, 3, 1, 64, 96),           # the narrow (4-quad, 8 x 32) tile
    ('dw', 672, 28, 5, 2, 40, 40),         # stride 2: the slice kernel (all channels in one workgroup per slice)
    ('dw', 240, 10, 3, 2, 31, 33),         # stride 2, odd sizes
    ('mbconv', 24, 6, 3, 1, 40, 48),       # fused expand + depthwise (Cexp 144: 5 chunks, the last of 16 channels)
    ('mbconv', 16, 4, 3, 2, 64, 64),       # ... stride 2
    ('stem', 32, 8, 3, 1, 96, 64),         # stem + block 0's depthwise conv
])
def test_se_gate_inside_depthwise_launch(dev, kind, C, Cse, k, s, H, W):
    """The squeeze-excite gate finished INSIDE the launch that produces the depthwise output (csrc/se_tail.h; `se=` of
    ops.dwconv / mbconv_expand_dw / stem_dw: every workgroup adds its share of W1 . sums, the last workgroup of an image
    finishes) == the gate of the separate launch (squeeze sums + mydet_se_gate_f32) to float32 round-off and == float64 of
    the launch's own output; the output map is bit-identical with and without the tail; three runs give the same bits (the
    share hand-over leaves every image's launch counter one step further, the sums run in a fixed order); batch of 3."""
    from mydetection_amd import ops
    from mydetection_amd.external.efficientnet.model import static_same_pad
    g = torch.Generator().manual_seed(C * 7 + H)
    B = 3
    pad = static_same_pad(k, s, 240)
    if kind == 'dw':
        x = torch.randn(B, C, H, W, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
        wd = (torch.randn(k, k, C, generator=g) / k).to(dev)
        sc, sh = (torch.rand(C, generator=g) + 0.5).to(dev), (torch.randn(C, generator=g) * 0.3).to(dev)
        Cg = C

        def run(se=None):
            if se is None:
                return ops.dwconv(x, wd, sc, sh, k, s, pad, ops.ACT_SWISH, squeeze=True)
            return ops.dwconv(x, wd, sc, sh, k, s, pad, ops.ACT_SWISH, se=se)
    elif kind == 'mbconv':
        cin, Cg = C, C * 6
        x = torch.randn(B, cin, H, W, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
        we = (torch.randn(Cg, 1, 1, cin, generator=g) / cin ** 0.5).to(dev)
        wd = (torch.randn(k, k, Cg, generator=g) / k).to(dev)
        sh0, sh1 = (torch.randn(Cg, generator=g) * 0.3).to(dev), (torch.randn(Cg, generator=g) * 0.3).to(dev)

        def run(se=None):
            return ops.mbconv_expand_dw(x, we, sh0, wd, sh1, k, s, pad, se=se)
    else:
        Cg = 32
        x = (torch.rand(B, 3, H, W, generator=g) * 2 - 1).to(dev)
        ws = (torch.randn(32, 3, 3, 3, generator=g) * 0.3).to(dev)
        wd = (torch.randn(3, 3, 32, generator=g) * 0.3).to(dev)
        sh0, sh1 = (torch.randn(32, generator=g) * 0.2).to(dev), (torch.randn(32, generator=g) * 0.2).to(dev)

        def run(se=None):
            return ops.stem_dw(x, ws, sh0, wd, sh1, (0, 0, 1, 1), se=se)
    w1 = (torch.randn(Cse, Cg, generator=g) / Cg ** 0.5).to(dev)
    b1 = (torch.randn(Cse, generator=g) * 0.1).to(dev)
    w2t = (torch.randn(Cse, Cg, generator=g) * 0.3).to(dev)
    b2 = (torch.randn(Cg, generator=g) * 0.1).to(dev)
    y0, partial = run()
    gate0 = ops.se_gate(partial, y0.shape[2] * y0.shape[3], w1, b1, w2t, b2)
    y1, gate1 = run((w1, b1, w2t, b2))
    assert torch.equal(y1.contiguous(), y0.contiguous())
    assert tuple(gate1.shape) == (B, Cg)
    m = y0.double().mean(dim=(2, 3)).cpu()
    h = m @ w1.double().cpu().t() + b1.double().cpu()
    h = h * torch.sigmoid(h)
    ref = torch.sigmoid(h @ w2t.double().cpu() + b2.double().cpu())
    assert (gate1.cpu().double() - ref).abs().max().item() < 3e-6
    assert (gate1 - gate0).abs().max().item() < 3e-6
    for _ in range(2):
        y2, gate2 = run((w1, b1, w2t, b2))
        assert torch.equal(gate2, gate1) and torch.equal(y2.contiguous(), y0.contiguous())
    from mydetection_amd import _lib
    hdr = ops.se_shares(dev, 1).view(torch.int32)[:_lib.SE_EPOCH_WORDS]
    assert int(hdr[0]) > 1 and int(hdr[1]) == 0 and not bool(hdr[2:].any())      # one step of the launch counter per launch
    # the launch counter wraps past 2^32 - 1 to 1 (0 is the tag of a fresh buffer): same gate before, across and after the wrap
    before = int(hdr[0])
    hdr[0] = -2                                              # 0xFFFFFFFE
    for want in (-1, 1, 2):
        _, gate3 = run((w1, b1, w2t, b2))
        assert torch.equal(gate3, gate1) and int(hdr[0]) == want and int(hdr[1]) == 0, want
    hdr[0] = before + 3
    # a launch with another batch size in between (the epoch is the buffer's, not an image slot's)
    if kind == 'dw':
        y4, gate4 = ops.dwconv(x[:2], wd, sc, sh, k, s, pad, ops.ACT_SWISH, se=(w1, b1, w2t, b2))
        assert torch.equal(gate4, gate1[:2]) and int(hdr[0]) == before + 4
        _, gate5 = run((w1, b1, w2t, b2))
        assert torch.equal(gate5, gate1)


@pytest.mark.parametrize('Cin,Cout,H,W,gated,res,act', [
    (16, 16, 192, 192, True, True, 0), (32, 16, 181, 183, True, False, 0), (96, 24, 192, 176, True, False, 0),
    (144, 24, 181, 183, True, True, 0), (144, 40, 192, 176, True, False, 0), (240, 40, 181, 183, True, True, 0),
    (32, 44, 192, 176, False, True, 2), (16, 8, 200, 168, False, False, 1), (144, 48, 181, 183, False, False, 0),
    # K not a multiple of 16 (masked last chunk) and output channels cut into slabs
    (40, 240, 81, 83, False, False, 2), (112, 672, 41, 39, False, False, 2), (80, 480, 40, 40, False, True, 2),
    (192, 1152, 20, 20, False, False, 2), (88, 88, 37, 41, False, False, 0), (24, 144, 90, 77, False, False, 1),
    (40, 40, 181, 183, True, True, 0), (4, 12, 64, 64, False, False, 0), (236, 100, 33, 31, False, True, 0)])
def test_pointwise_skinny(dev, monkeypatch, Cin, Cout, H, W, gated, res, act):
    """The LDS-free skinny 1x1 kernel (csrc/pointwise.hip; mydet_conv2d_igemm_f32 routes Cout <= 48, Cin in
    {16, 32, 96, 144, 240}, >= 65 536 pixels to it): against a float64 conv, with the SE gate on x, BatchNorm terms,
    activation, residual, ragged pixel counts (a last workgroup past the end, pixel blocks that straddle the two images)
    and padded leading dimensions; and against the tiled kernel (MYDET_PW_SKINNY=0) to float32 rounding."""
    from mydetection_amd import ops
    g = torch.Generator().manual_seed(Cin * 100 + Cout)
    B = 2
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 1, 1, generator=g) / Cin ** 0.5
    scale, shift = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.1
    gate = torch.rand(B, Cin, generator=g) if gated else None
    r = torch.randn(B, Cout, H, W, generator=g) if res else None
    xin = x.double() * gate.double().view(B, Cin, 1, 1) if gated else x.double()
    ref = F.conv2d(xin, w.double()) * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)
    if act == 1:
        ref = torch.where(ref > 0, ref, ref * 0.1)
    if act == 2:
        ref = ref * torch.sigmoid(ref)
    if res:
        ref = ref + r.double()
    xd = x.to(dev).contiguous(memory_format=torch.channels_last)
    args = (xd, w.permute(0, 2, 3, 1).contiguous().to(dev), scale.to(dev), shift.to(dev), 1, 1, (0, 0, 0, 0), act)
    kw = dict(residual=r.to(dev).contiguous(memory_format=torch.channels_last) if res else None,
              gate=gate.to(dev) if gated else None)
    monkeypatch.setenv('MYDET_PW_WIDE', '1')              # every shape of this test through pointwise.hip
    y = ops.conv2d(*args, **kw)
    tol = 2e-5 * max(1.0, ref.abs().max().item())
    assert (y.cpu().double() - ref).abs().max() < tol
    y_pad = ops.conv2d(*args, out_ld=Cout + 8, **kw)                  # a leading dimension wider than Cout
    assert torch.equal(y_pad.contiguous(), y.contiguous())
    monkeypatch.setenv('MYDET_PW_SKINNY', '0')
    y_tiled = ops.conv2d(*args, **kw)
    monkeypatch.delenv('MYDET_PW_SKINNY')
    assert (y_tiled.cpu().double() - ref).abs().max() < tol
    assert not torch.equal(y_tiled, y) or Cin == 16       # two kernels, two summation orders (K = 16 can coincide)


def test_maxpool_and_bifpn_fuse(dev):
    from mydetection_amd import ops
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 88, 10, 10, generator=g)
    y = ops.maxpool3s2(x.to(dev))
    assert torch.equal(y.cpu(), F.max_pool2d(x, 3, 2, 1))
    x2 = torch.randn(2, 24, 7, 9, generator=g)                       # odd sizes
    assert torch.equal(ops.maxpool3s2(x2.to(dev)).cpu(), F.max_pool2d(x2, 3, 2, 1))
    swish = lambda t: t * torch.sigmoid(t)                           # noqa: E731
    a, b_half, c_dbl = torch.randn(2, 88, 8, 8, generator=g), torch.randn(2, 88, 4, 4, generator=g), torch.randn(2, 88, 16, 16, generator=g)
    for wts in (torch.tensor([0.7, 1.2]), torch.tensor([-0.5, 1.0])):
        w = F.relu(wts)
        w = w / (w.sum() + 0.0001)
        ref = swish(sum([wi * f for wi, f in zip(w, [a, F.interpolate(b_half, scale_factor=(2, 2), mode='nearest')])]))
        out = ops.bifpn_fuse([a.to(dev), b_half.to(dev)], [ops.FUSE_SAME, ops.FUSE_UP2X], wts.to(dev))
        np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), rtol=2e-6, atol=1e-7)
    wts = torch.tensor([0.9, 0.4, 1.3])
    w = F.relu(wts)
    w = w / (w.sum() + 0.0001)
    a2 = torch.randn(2, 88, 8, 8, generator=g)
    ref = swish(sum([wi * f for wi, f in zip(w, [a, a2, F.max_pool2d(c_dbl, 3, 2, 1)])]))
    out = ops.bifpn_fuse([a.to(dev), a2.to(dev), c_dbl.to(dev)], [ops.FUSE_SAME, ops.FUSE_SAME, ops.FUSE_POOL], wts.to(dev))
    np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), rtol=2e-6, atol=1e-7)


@pytest.mark.parametrize('B,Cin,Cout,k,HW', [(11, 1024, 128, 1, 80),     # 64x64 tiles: 2200 = 2 rounds of 1024 + 152
                                              (69, 128, 256, 3, 32)])     # 128x128 tiles: 1104 = 2 rounds of 512 + 80
def test_conv_igemm_split_k_tail(dev, B, Cin, Cout, k, HW):
    """Grids of R full rounds + a small remainder run the remainder tiles split along K (partials in the
    workspace, summed in fixed order by the fixup launch); result must match a float32 CPU conv."""
    from mydetection_amd import ops
    g = torch.Generator().manual_seed(8)
    x = torch.randn(B, Cin, HW, HW, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    scale, shift = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.1
    res = torch.randn(B, Cout, HW, HW, generator=g)
    p = (k - 1) // 2
    ref = F.leaky_relu(F.conv2d(x, w, None, 1, p) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1), 0.1) + res
    args = (x.to(dev).contiguous(memory_format=torch.channels_last), w.permute(0, 2, 3, 1).contiguous().to(dev),
            scale.to(dev), shift.to(dev), k, 1, (p, p, p, p), 1)
    y = ops.conv2d(*args, residual=res.to(dev).contiguous(memory_format=torch.channels_last))
    assert (y.cpu() - ref).abs().max().item() <= 1e-4 * max(1.0, ref.abs().max().item())
    y2 = ops.conv2d(*args, residual=res.to(dev).contiguous(memory_format=torch.channels_last))
    assert torch.equal(y, y2)                                  # deterministic


def test_post_process_randomised_sweep_vs_oracle(dev):
    """40 random workloads (candidate count 0..30 000, 1..90 classes, clustered boxes so that many IoUs sit near the
    threshold, quantised scores so that ties are common, thresholds drawn at random): kept indices, order and
    count equal the oracle's bit for bit."""
    from mydetection_amd import ops
    from oracle import postprocess as pp
    rng = np.random.Generator(np.random.PCG64(2024))
    for trial in range(40):
        N = int(rng.choice([0, 1, 7, 513, 2000, 9000, 30000]))
        B = int(rng.integers(1, 4))
        n_cls = int(rng.choice([1, 3, 20, 90]))
        n_clusters = int(rng.integers(1, 40))
        centres = rng.random((n_clusters, 2), dtype=np.float32) * 600
        which = rng.integers(0, n_clusters, size=(B, N))
        b = np.empty((B, N, 4), np.float32)
        b[..., :2] = centres[which] + rng.normal(0, 6, size=(B, N, 2)).astype(np.float32)
        b[..., 2:] = rng.choice(np.array([24, 32, 48], np.float32), size=(B, N, 2)) + rng.normal(0, 2, size=(B, N, 2)).astype(np.float32)
        b[..., 2:] = np.abs(b[..., 2:])
        c = rng.integers(0, n_cls, size=(B, N)).astype(np.int64)
        s = rng.random((B, N), dtype=np.float32)
        if trial % 3 == 0:
            s = np.round(s * 64) / 64                       # heavy score ties
        conf = float(rng.choice([0.0, 0.005, 0.3, 0.9]))
        thr = float(rng.choice([0.0, 0.3, 0.45, 0.5, 0.7, 1.0]))
        rec = ops.postprocess(torch.from_numpy(b).to(dev), torch.from_numpy(c).to(dev), torch.from_numpy(s.astype(np.float32)).to(dev),
                              conf, thr)
        for i in range(B):
            ob, oc, os_, src = pp.post_process(b[i], c[i], s[i].astype(np.float32), conf, thr)
            k = int(rec['count'][i])
            assert k == len(src), (trial, i, N, n_cls, conf, thr, k, len(src))
            np.testing.assert_array_equal(rec['index'][i, :k].cpu().numpy().astype(np.int64), src)
            np.testing.assert_array_equal(rec['score'][i, :k].cpu().numpy(), os_)


def _sepconv_ref(inputs, modes, fuse_w, w_dw, w_pw, scale, shift, act):
    """float64 restatement of one pyramid node: [fusion + swish ->] depthwise 3x3 -> pointwise (+BN fold, act)."""
    ins = []
    for t, m in zip(inputs, modes):
        t = t.double()
        if m == 1:
            t = F.interpolate(t, scale_factor=(2, 2), mode='nearest')
        elif m == 2:
            t = F.max_pool2d(t, 3, 2, 1)
        ins.append(t)
    if len(ins) > 1:
        w = F.relu(fuse_w.double())
        w = w / (w.sum() + 0.0001)
        x = sum([wi * f for wi, f in zip(w, ins)])
        x = x * torch.sigmoid(x)
    else:
        x = ins[0]
    C = x.shape[1]
    y = F.conv2d(x, w_dw.double().permute(2, 0, 1).reshape(C, 1, 3, 3), None, 1, 1, 1, C)
    y = F.conv2d(y, w_pw.double().reshape(w_pw.shape[0], C, 1, 1))
    y = y * (scale.double().view(1, -1, 1, 1) if scale is not None else 1.0) + shift.double().view(1, -1, 1, 1)
    return y * torch.sigmoid(y) if act == 2 else y


def test_sepconv_nodes_vs_fp64(dev):
    """The fused pyramid node (fusion + swish -> depthwise 3x3 -> pointwise on FP32 MFMA -> BN/act) against float64:
    every pre-stage (identity, 2-input with nearest-2x, 3-input with the 3x3/2 max pool), odd / partial tiles (5x5,
    10x10, 13x7), all Cout of the D1 family (88, 720, 36, 4), with and without BatchNorm, several nodes per launch --
    and against the three-launch path it replaces."""
    from mydetection_amd import ops
    g = torch.Generator().manual_seed(17)
    C, B = 88, 3

    def node(hw, n_in, cout, act, bn, modes=None):
        H, W = hw
        modes = modes or [0] * n_in
        shapes = {0: (H, W), 1: (H // 2, W // 2), 2: (H * 2, W * 2)}
        inputs = [torch.randn(B, C, *shapes[m], generator=g) for m in modes]
        return dict(inputs=inputs, modes=modes, fuse_w=torch.tensor([0.8, 1.3, -0.4][:n_in]) if n_in > 1 else None,
                    w_dw=torch.randn(3, 3, C, generator=g) / 3.0, w_pw=torch.randn(cout, C, generator=g) / C ** 0.5,
                    scale=torch.rand(cout, generator=g) + 0.5 if bn else None, shift=torch.randn(cout, generator=g) * 0.2,
                    cout=cout, act=act)

    groups = [
        [node((16, 16), 1, 88, 2, True), node((5, 5), 1, 88, 2, True), node((10, 10), 1, 88, 2, True), node((13, 7), 1, 88, 2, True)],
        [node((8, 8), 2, 88, 0, True, [0, 1]), node((10, 10), 3, 88, 0, True, [0, 0, 2]), node((20, 20), 2, 88, 0, True, [0, 2])],
        [node((9, 9), 1, 720, 0, False), node((9, 9), 1, 36, 0, False), node((6, 5), 1, 4, 0, False)],
    ]
    for grp in groups:
        dev_nodes = [dict(inputs=[t.to(dev).contiguous(memory_format=torch.channels_last) for t in nd['inputs']], modes=nd['modes'],
                          fuse_weights=nd['fuse_w'].to(dev) if nd['fuse_w'] is not None else None,
                          w_dw=nd['w_dw'].to(dev), w_pw=ops.pack_pointwise(nd['w_pw'].to(dev)),
                          scale=nd['scale'].to(dev) if nd['scale'] is not None else None, shift=nd['shift'].to(dev),
                          cout=nd['cout'], act=nd['act']) for nd in grp]
        outs = ops.sepconv_nodes(dev_nodes)
        again = ops.sepconv_nodes(dev_nodes)
        for nd, dn, y, y2 in zip(grp, dev_nodes, outs, again):
            ref = _sepconv_ref(nd['inputs'], nd['modes'], nd['fuse_w'], nd['w_dw'], nd['w_pw'], nd['scale'], nd['shift'], nd['act'])
            assert y.shape == ref.shape
            err = (y.cpu().double() - ref).abs().max().item()
            assert err <= 2e-5 * max(1.0, ref.abs().max().item()), (nd['cout'], nd['modes'], err)
            assert torch.equal(y, y2)
            # the three launches it replaces
            x = dn['inputs'][0] if len(dn['inputs']) == 1 else ops.bifpn_fuse(dn['inputs'], dn['modes'], dn['fuse_weights'])
            t = ops.dwconv(x, dn['w_dw'], None, None, 3, 1, (1, 1, 1, 1), ops.ACT_NONE)
            old = ops.conv2d(t, nd['w_pw'].reshape(nd['cout'], 1, 1, C).to(dev), dn['scale'], dn['shift'], 1, 1, (0, 0, 0, 0), nd['act'])
            assert (y - old).abs().max().item() <= 1e-5 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize('k,s,cin,hw', [(3, 2, 16, (40, 48)), (3, 1, 24, (24, 32)), (5, 2, 24, (22, 26)), (5, 1, 40, (17, 23)),
                                        (3, 2, 40, (16, 16)), (3, 1, 24, (8, 16))])
def test_mbconv_expand_dw_vs_fp64(dev, k, s, cin, hw):
    """Fused expand 1x1 + BN + swish -> depthwise k x k (static-SAME pads of the expanded map) + BN + swish + SE squeeze sums
    against float64, and against the two launches it replaces (expand conv, depthwise with fused squeeze)."""
    from mydetection_amd import ops
    from mydetection_amd.external.efficientnet.model import static_same_pad
    g = torch.Generator().manual_seed(100 * k + 10 * s + cin)
    B, (H, W), cexp = 3, hw, cin * 6
    x = torch.randn(B, cin, H, W, generator=g)
    we = torch.randn(cexp, cin, generator=g) / cin ** 0.5
    wd = torch.randn(k, k, cexp, generator=g) / k
    sc0, sh0 = torch.rand(cexp, generator=g) + 0.5, torch.randn(cexp, generator=g) * 0.3
    sc1, sh1 = torch.rand(cexp, generator=g) + 0.5, torch.randn(cexp, generator=g) * 0.3
    pad = static_same_pad(k, s, 240)                                  # (top, left, bottom, right)
    e = F.conv2d(x.double(), we.double().view(cexp, cin, 1, 1)) * sc0.double().view(1, -1, 1, 1) + sh0.double().view(1, -1, 1, 1)
    e = e * torch.sigmoid(e)
    e = F.pad(e, (pad[1], pad[3], pad[0], pad[2]))
    y = F.conv2d(e, wd.double().permute(2, 0, 1).reshape(cexp, 1, k, k), None, s, 0, 1, cexp)
    y = y * sc1.double().view(1, -1, 1, 1) + sh1.double().view(1, -1, 1, 1)
    ref = y * torch.sigmoid(y)
    args = [t.to(dev) for t in (we.view(cexp, 1, 1, cin).contiguous(), sc0, sh0, wd, sc1, sh1)]
    xd = x.to(dev).contiguous(memory_format=torch.channels_last)
    fused_args = (ops.fold_scale(args[0], args[1]), args[2], ops.fold_scale(args[3], args[4]), args[5])
    out, partial = ops.mbconv_expand_dw(xd, *fused_args, k, s, pad)
    assert out.shape == ref.shape
    tol = 2e-5 * max(1.0, ref.abs().max().item())
    assert (out.cpu().double() - ref).abs().max().item() <= tol
    sums = partial[:, :-1].sum(dim=1).cpu().double()                 # per-tile sums -> per-image channel sums
    np.testing.assert_allclose(sums.numpy(), ref.sum(dim=(2, 3)).numpy(), rtol=1e-4, atol=1e-3)
    out2, partial2 = ops.mbconv_expand_dw(xd, *fused_args, k, s, pad)
    assert torch.equal(out, out2) and torch.equal(partial[:, :-1], partial2[:, :-1])
    # the two launches it replaces
    ex = ops.conv2d(xd, args[0], args[1], args[2], 1, 1, (0, 0, 0, 0), ops.ACT_SWISH)
    old, _ = ops.dwconv(ex, args[3], args[4], args[5], k, s, pad, ops.ACT_SWISH, squeeze=True)
    assert (out - old).abs().max().item() <= 1e-5 * max(1.0, ref.abs().max().item())


def test_last_conv_padded_output_channels(dev):
    """The FCOS head's dense 3x3 last conv (88 -> 81 = conf + 80 classes, reference models/rpns.py:245-266) is computed
    with 84 output channels (zero weight rows) so that the Winograd kernels take it: the 81-channel view must equal
    the float64 convolution, and the three padding channels must hold exactly the bias-free zeros."""
    from mydetection_amd.models.rpns import _LastConv
    torch.manual_seed(3)
    m = _LastConv(88, 81, 3, 1, padding=1).to(dev).eval()
    with torch.no_grad():
        m.weight.normal_(0, 0.05)
        m.bias.normal_(0, 0.5)
    for B, H in ((2, 20), (32, 80)):                     # F(2x2) path / F(4x4) path (1200 workgroups)
        x = torch.randn(B, 88, H, H, device=dev).contiguous(memory_format=torch.channels_last)
        with torch.no_grad():
            y = m(x)
        assert tuple(y.shape) == (B, 81, H, H) and y.stride(1) == 1 and y.stride(3) == 84
        ref = F.conv2d(x.double().cpu(), m.weight.double().cpu(), m.bias.double().cpu(), padding=1)
        assert (y.cpu().double() - ref).abs().max().item() <= 6e-5 * max(1.0, ref.abs().max().item())
        pad = torch.as_strided(y, (B, 3, H, H), y.stride(), y.storage_offset() + 81)
        assert torch.equal(pad, torch.zeros_like(pad))

