"""CPU-only: host logic of the drop-in surface, and that the C-ABI library loads and exports every
symbol include/mydet.h declares (no compute without a GPU)."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from mydetection_amd import _lib
    header = open(os.path.join(ROOT, 'include', 'mydet.h')).read()
    declared = set(re.findall(r'\b(?:int|int64_t)\s+(mydet_\w+)\s*\(', header))
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))
    handle = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(handle, name), name
    assert _lib.lib().mydet_abi_version() == _lib.ABI_VERSION == 2
    assert f'#define MYDET_ABI_VERSION {_lib.ABI_VERSION} ' in open(os.path.join(ROOT, 'include', 'mydet.h')).read()


def test_argument_errors_are_reported_before_launch():
    from mydetection_amd import _lib
    lib = _lib.lib()
    null = ctypes.c_void_p(0)
    assert lib.mydet_conv2d_igemm_f32(null, 0, null, null, null, null, 0, null, null, 0, null, 0, 1, 1, 1, 4, 4, 1, 1, 1, 0, 0,
                                      1, 1, 0, null) == -1
    assert lib.mydet_postprocess_f32(null, null, null, 1, 1 << 20, 0.5, 0.5, 512, null, null, null, null, null, null,
                                     null) == -2
    assert lib.mydet_postprocess_records_f32(null, null, null, 1, 1 << 20, 0.5, 0.5, null, null, null) == -2
    assert lib.mydet_postprocess_records_f32(null, null, null, 1, 100, 0.5, 0.5, null, null, null) == -1
    header = open(os.path.join(ROOT, 'include', 'mydet.h')).read()
    words = {m.group(1): m.group(2) for m in re.finditer(r'#define MYDET_REC_(\w+)\s+(.+)', header)}
    assert int(words['TOPK']) == _lib.REC_TOPK and int(words['BBOX']) == _lib.REC_BBOX and _lib.REC_WORDS == 4100
    assert lib.mydet_conv2d_wino_f32(null, 0, null, null, null, null, 0, null, 0, null, 0, 1, 8, 8, 8, 8, 0, null) == -1
    assert lib.mydet_wino_weights_floats(64, 12) == 0 and lib.mydet_wino_weights_floats(70, 16) == 16 * 16 * 128
    with pytest.raises(_lib.MydetError):
        _lib.check(-1, 'x')


def test_missing_library_fails_loudly(monkeypatch):
    from mydetection_amd import _lib
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', '/nonexistent/libmydet_hip.so')
    with pytest.raises(_lib.MissingHipLibrary):
        _lib.lib()


def test_no_cpu_path():
    from mydetection_amd import ops
    from mydetection_amd.utils.structures import ImageObjects
    with pytest.raises(RuntimeError):
        ops.conv2d(torch.zeros(1, 4, 2, 2), torch.zeros(4, 1, 1, 4), None, None, 1, 1, (0, 0, 0, 0), 0)
    if not torch.cuda.is_available():
        d = ImageObjects(torch.zeros(2, 4), torch.zeros(2, dtype=torch.int64), None, torch.ones(2))
        with pytest.raises(RuntimeError):
            d.post_process(0.5, 0.5)


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'mydetection_amd')):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, re.M), f
                assert not re.search(r'import_module\([^)]*oracle|__import__\([^)]*oracle', src), f
                for needle in ('oracle/_build', 'oracle/_ref', 'libnms_ref', 'nms_ref_f32'):   # the checker's binaries
                    assert needle not in src, (f, needle)


def test_registry_contract_and_state_dict_keys():
    from mydetection_amd.models import registry
    from mydetection_amd.models.general import load_config, state_dict_template
    cfg = load_config('yolov3_80')
    assert cfg['model.backbone.out_channels'] == 'PLACEHOLDER'
    with torch.device('meta'):
        registry.get_backbone(cfg)
        assert cfg['model.backbone.out_channels'] == (256, 512, 1024) and cfg['model.backbone.out_strides'] == (8, 16, 32)
        registry.get_fpn(cfg)
        assert cfg['model.fpn.out_channels'] == (256, 512, 1024) and cfg['model.fpn.out_strides'] == (8, 16, 32)
        registry.get_rpn(cfg)
    assert registry.get_det_layer(cfg).__name__ == 'YOLOLayer'
    for fn, key in ((registry.get_backbone, 'model.backbone.name'), (registry.get_fpn, 'model.fpn.name')):
        with pytest.raises(Exception, match='Unknown'):
            fn({**cfg, key: 'nope'})
    with pytest.raises(NotImplementedError):
        registry.get_rpn({**cfg, 'model.rpn.name': 'nope'})
    with pytest.raises(NotImplementedError):
        registry.get_det_layer({**cfg, 'model.pred_layer': 'nope'})
    sd = state_dict_template('yolov3_80')
    assert len(sd) == 438
    assert sd['backbone.netlist.0.conv.weight'].shape == (32, 3, 3, 3)
    assert sd['backbone.netlist.28.cbl_1.bn.running_var'].shape == (1024,)
    assert sd['fpn.branch_P3.process.conv.weight'].shape == (128, 256, 1, 1)
    assert sd['fpn.branch_P4.cbl_0.conv.weight'].shape == (256, 768, 1, 1)
    assert sd['rpn.heads.conv_2.weight'].shape == (255, 1024, 1, 1) and sd['rpn.heads.conv_0.bias'].shape == (255,)
    n_params = sum(v.numel() for k, v in sd.items() if not k.endswith(('running_mean', 'running_var', 'num_batches_tracked')))
    assert n_params == 61949149


def test_ultralytics_plugins_contract():
    """get_backbone('ultralytics') / get_fpn('ultralytics') (reference: models/registry.py:25-28,53-57): derived cfg keys,
    the reference's 462 state_dict keys for u5m_yv3 with their shapes (checked key by key against the imported
    reference when the fixtures were made), the quirks kept (FPN depth scaled by the CHANNEL multiple)."""
    from mydetection_amd.models import registry
    from mydetection_amd.models.general import load_config, state_dict_template
    cfg = load_config('u5m_yv3')
    with torch.device('meta'):
        bb = registry.get_backbone(cfg)
        assert cfg['model.backbone.out_channels'] == [192, 384, 768] and cfg['model.backbone.out_strides'] == [8, 16, 32]
        fpn = registry.get_fpn(cfg)
        assert cfg['model.fpn.out_channels'] == [192, 384, 768]
    assert len(bb.netlist) == 10 and len(bb.netlist[2]) == 2 and len(bb.netlist[4].m) == 6 and len(bb.netlist[9].m) == 4
    assert len(fpn.to_p5.m) == 2 and not fpn.to_p5.m[0].add and bb.netlist[4].m[0].add
    sd = state_dict_template('u5m_yv3')
    assert len(sd) == 462
    assert sd['backbone.netlist.0.conv.conv.weight'].shape == (48, 12, 3, 3)
    assert sd['backbone.netlist.4.cv2.weight'].shape == (96, 192, 1, 1) and sd['backbone.netlist.4.bn.weight'].shape == (192,)
    assert sd['backbone.netlist.8.cv2.conv.weight'].shape == (768, 1536, 1, 1)
    assert sd['fpn.to_p4.0.conv.weight'].shape == (384, 1152, 1, 1) and sd['fpn.to_p3.1.cv4.bn.running_var'].shape == (192,)
    assert sd['rpn.heads.conv_2.weight'].shape == (255, 768, 1, 1)
    sd2 = state_dict_template('u5m_fcs2')
    assert sd2['rpn.heads.conv_0.weight'].shape == (85, 192, 1, 1) and len(sd2) == 462
    with pytest.raises(NotImplementedError):
        registry.get_backbone({**load_config('u5m_yv3'), 'model.ultralytics.first': 'Conv2d'})


def test_image_objects_host_logic():
    from mydetection_amd.utils.structures import ImageObjects
    b = torch.tensor([[10., 10., 4., 4.], [20., 20., 6., 2.], [5., 5., 1., 1.]])
    d = ImageObjects(b, torch.tensor([3, 1, 3]), None, torch.tensor([0.25, 0.9, 0.5]), 'cxcywh', (32, 32))
    assert len(d) == 3 and len(d[1]) == 1 and len(d[torch.tensor([True, False, True])]) == 2
    d.sort_by_score_()
    assert d.cats.tolist() == [1, 3, 3]
    d.category_filter_([3])
    assert d.scores.tolist() == [0.5, 0.25]
    if not torch.cuda.is_available():           # the json numbers come from a HIP launch: loud failure without the GPU
        with pytest.raises(RuntimeError):
            d.to_json(img_id='a')
    from mydetection_amd.utils.structures import _category_table, _json_rows
    assert _category_table({0: 'a'}, 'cpu') == (None, {0: 'a'})
    assert _json_rows([[4.5, 4.5, 1.0, 1.0, 0.5]], [4], 'a', None) == [{'image_id': 'a', 'category_id': 4, 'bbox': [4.5, 4.5, 1.0, 1.0], 'score': 0.5}]
    with pytest.raises(AssertionError):
        ImageObjects(b, torch.tensor([1, 2, 3], dtype=torch.int32))
    with pytest.raises(NotImplementedError):
        ImageObjects(b, torch.tensor([1, 2, 3]), bb_format='x1y1x2y2')


def test_model_rejects_training_inputs():
    from mydetection_amd.models.general import OneStageBBox, load_config
    with torch.device('meta'):
        m = OneStageBBox(load_config('yolov3_80'))
    assert m.input_format == 'RGB_1' and m.bb_format == 'cxcywh'
    with pytest.raises(NotImplementedError):
        m(torch.zeros(1, 3, 32, 32), labels=[])


def test_preprocess_shapes():
    import numpy as np
    import PIL.Image
    from mydetection_amd.utils import image_ops
    img = PIL.Image.fromarray(np.zeros((300, 400, 3), np.uint8))
    r = image_ops.resize_pil(img, 320, shorter=False)
    assert (r.height, r.width) == (240, 320)
    p = image_ops.pad_to_divisible(r, 32)
    assert (p.height, p.width) == (256, 320)
    sq, _, info = image_ops.rect_to_square(img, None, 256)
    assert sq.size == (256, 256) and info == (400, 300, 0, 32, 256, 192)
    t = image_ops.format_tensor_img(image_ops.to_tensor(p), 'RGB_1_norm')
    assert t.shape == (3, 256, 320)


def test_pil_resize_restatement_and_tables():
    """oracle/pil_resize.py (numpy restatement of Pillow's 8-bit bilinear resize) equals PIL itself bit for bit, and the
    product's vectorised coefficient tables (utils/image_ops.resample_tables, uploaded to the device resize kernel)
    equal the restatement's."""
    import numpy as np
    import PIL.Image
    from mydetection_amd.utils.image_ops import resample_tables
    from oracle import pil_resize
    rng = np.random.default_rng(3)
    for t in range(12):
        H, W = int(rng.integers(5, 120)), int(rng.integers(5, 120))
        oh, ow = int(rng.integers(3, 140)), int(rng.integers(3, 140))
        if t % 4 == 0:
            oh = H
        if t % 5 == 0:
            ow = W
        img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        ref = np.array(PIL.Image.fromarray(img).resize((ow, oh), PIL.Image.BILINEAR))
        assert np.array_equal(pil_resize.resize_bilinear_u8(img, (oh, ow)), ref), ((H, W), (oh, ow))
    for n_in, n_out in ((640, 480), (480, 640), (37, 301), (1000, 64), (64, 1000), (7, 7), (2, 3)):
        b0, k0 = pil_resize.coefficients(n_in, n_out)
        b1, k1 = resample_tables(n_in, n_out)
        assert np.array_equal(b0, b1) and np.array_equal(k0, k1), (n_in, n_out)


def test_detector_geometry_matches_reference_golden(golden):
    """Detector._geometry (target size, offsets, input size, pad_info) against what the reference's _preprocess_pil
    produced for the fixture images; the PIL host path of utils/image_ops reproduces the reference tensor too."""
    import numpy as np
    import PIL.Image
    from mydetection_amd.api.detection import Detector
    from mydetection_amd.utils import image_ops
    g = golden('preprocess_json')
    det = Detector.__new__(Detector)
    for i in range(int(g['n_cases'])):
        mode, fmt, size, div = str(g[f'c{i}_mode']), str(g[f'c{i}_fmt']), int(g[f'c{i}_size']) or None, int(g[f'c{i}_div'])
        img, ref = g[f'c{i}_image'], g[f'c{i}_tensor']
        det.divisibe = div
        target, (top, left), out_hw, pad_info = det._geometry(img.shape[0], img.shape[1], mode, size)
        assert tuple(out_hw) == ref.shape[1:]
        want = g[f'c{i}_pad_info']
        assert (pad_info is None and want.size == 0) or np.array_equal(np.array(pad_info, dtype=np.float64), want)
        pil = PIL.Image.fromarray(img)
        if target is not None:
            pil = image_ops._resize(pil, target)
        canvas = np.zeros((out_hw[0], out_hw[1], 3), np.uint8)
        canvas[top:top + pil.height, left:left + pil.width] = np.array(pil)
        t = image_ops.format_tensor_img(image_ops.to_tensor(PIL.Image.fromarray(canvas)), fmt)
        assert np.array_equal(t.numpy(), ref), (i, mode)


def test_graph_cache_policy_and_workspace_refs():
    """Host bookkeeping behind the captured hipGraphs (no GPU call): the LRU moves hits to the young end, an eviction
    forgets the evicted key's sightings and doubles the sightings a key needs before it is captured (a workload cycling
    through more shapes than the cache holds stops capturing), stale graphs are dropped on lookup; a superseded
    F(4x4) workspace stays alive for whoever holds `ops.live_workspaces`."""
    import torch
    from mydetection_amd import ops
    from mydetection_amd.graph import GraphCache

    class G:
        def __init__(self):
            self.old = False

        def stale(self):
            return self.old

    c = GraphCache(capacity=2)
    assert not c.should_capture('a')
    c.note_eager('a')
    assert c.should_capture('a')
    ga = c.insert('a', G())
    c.note_eager('b')
    gb = c.insert('b', G())
    assert c.lookup('a') is ga                      # hit: 'a' becomes the youngest
    c.note_eager('c')
    c.insert('c', G())                              # evicts 'b' (the oldest), not 'a'
    assert c.lookup('b') is None and c.lookup('a') is ga and c.evictions == 1 and c.need == 2
    assert 'b' not in c.seen
    c.note_eager('b')
    assert not c.should_capture('b')                # needs two sightings now
    c.note_eager('b')
    assert c.should_capture('b')
    # cycling through 5 keys with room for 2: captures die out instead of happening on every other call
    c = GraphCache(capacity=2)
    for _ in range(200):
        for k in 'vwxyz':
            if c.lookup(k) is None:
                if c.should_capture(k):
                    c.insert(k, G())
                else:
                    c.note_eager(k)
    assert c.captures < 40, c.captures
    ga.old = True
    c2 = GraphCache(capacity=2)
    c2.insert('a', ga)
    assert c2.lookup('a') is None and 'a' not in c2.graphs
    del gb

    dev = torch.device('cpu')
    for k in [k for k in ops._WINO4_WS if k[:2] == (dev.type, dev.index)]:
        ops._WINO4_WS.pop(k)
    ws1 = ops.wino4_workspace(dev, 1024)
    held = ops.live_workspaces(dev)
    p1 = ws1.data_ptr()
    del ws1
    ws2 = ops.wino4_workspace(dev, 1 << 20)
    assert ws2.data_ptr() != p1 and any(t.data_ptr() == p1 for t in held)       # the old block is still owned by `held`
    assert ops.wino4_workspace(dev, 4096) is ws2                                 # no shrink, no churn
    # batch lanes (graph.GraphedPath): a lane has scratch of its own, and a graph holds every lane's
    with ops.lane(1):
        wl = ops.wino4_workspace(dev, 4096)
        cl = ops.conv_workspace(dev)
        with ops.lane(0):
            assert ops.wino4_workspace(dev, 4096) is ws2
        assert ops.wino4_workspace(dev, 16) is wl
    assert wl is not ws2 and cl is not ops.conv_workspace(dev) and ops.wino4_workspace(dev, 16) is ws2
    live = ops.live_workspaces(dev)
    assert all(any(t is w for t in live) for w in (ws2, wl, cl, ops.conv_workspace(dev)))
    for k in [k for k in ops._WINO4_WS if k[:2] == (dev.type, dev.index)]:
        ops._WINO4_WS.pop(k)
    with __import__('pytest').raises(ValueError):
        ops.check_counts([3, -1, 0])
    assert ops.check_counts([0, 5]) == [0, 5]


def test_conv_p3_patch_layout_is_conflict_free():
    """The LDS layout of conv_p3_kernel's input patch (csrc/conv_p3.hip: P3Geom, the position / sigma formulas in its header) against
    the ds_read_b128 lane groups and bank rule of MI355X_MICROARCH.md, replayed on the host (tools/r06/p3_layout_search.py): every
    A-fragment read of every tap, row block and lane group touches sixteen distinct 16-byte slots -- for both strides.  (On the GPU:
    SQ_LDS_BANK_CONFLICT = 0, profiles/r06_conv_p3.txt.)  The constants are read from the kernel source so that the two cannot drift."""
    import importlib.util
    import re
    spec = importlib.util.spec_from_file_location('p3_layout_search', os.path.join(ROOT, 'tools', 'r06', 'p3_layout_search.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    src = open(os.path.join(ROOT, 'mydetection_amd', 'csrc', 'conv_p3.hip')).read()
    geo = {(int(m.group(1)), int(m.group(2))): tuple(int(v) for v in m.group(3, 4, 5, 6, 7))
           for m in re.finditer(r'P3Geom<(\d), (\d)> \{ static constexpr int TH = (\d+), TWL = (\d+), PH = (\d+), ROWLEN = (\d+), PJ0 = (\d+); \}', src)}
    # (stride, shape) -> (tile rows, log2 tile columns, patch rows, positions per patch row, positions of the even columns)
    assert geo == {(2, 0): (8, 4, 17, 36, 17), (1, 0): (8, 4, 10, 24, 0), (2, 1): (16, 3, 33, 20, 9), (2, 2): (32, 2, 65, 9, 5)}, geo
    assert 'return SHAPE == 0 ? (j >> 3) & 1 : ((j >> 3) + (py >> 1)) & 1;' in src            # the swizzle bit, staging and fragment reads alike
    for (S, shape), (TH, TWL, PH, ROWLEN, PJ0) in geo.items():
        TW = 1 << TWL
        assert TH * TW == 128 and PH == S * (TH - 1) + 3
        sig = (lambda py, j: (j >> 3) & 1) if shape == 0 else (lambda py, j: ((j >> 3) + (py >> 1)) & 1)
        if S == 2:
            pos = lambda py, px, R=ROWLEN, P=PJ0: py * R + (px & 1) * P + (px >> 1)          # noqa: E731
            sg = lambda py, px, f=sig: f(py, px >> 1)                                       # noqa: E731
        else:
            pos = lambda py, px, R=ROWLEN: py * R + px                                      # noqa: E731
            sg = lambda py, px, f=sig: f(py, px)                                            # noqa: E731
        assert mod.worst_conflict(pos, sg, S, TH, TW) == 1, (S, shape)
        # the positions of one patch row are distinct and inside the row
        cols = S * (TW - 1) + 3
        assert len({pos(0, px) for px in range(cols)}) == cols and max(pos(0, px) for px in range(cols)) < ROWLEN
    # ... and the check can fail: the unswizzled stride-1 layout is conflicted
    assert mod.worst_conflict(lambda py, px: py * 24 + px, lambda py, px: 0, 1, 8, 16) > 1


def test_se_tail_share_buffer_size_is_validated():
    """mydet_se_tail carries the byte count of its share buffer and every entry point that takes an in-launch
    squeeze-excite tail rejects an undersized one BEFORE launching anything (MYDET_E_BADARG; no GPU needed: the pointers are
    never dereferenced on the host) -- 4 * (MYDET_SE_EPOCH_WORDS + 2 * B * groups * Cse) bytes."""
    import ctypes
    from mydetection_amd import _lib
    h = _lib.lib()
    B, C, Cse, K, Ho = 4, 1152, 48, 5, 20
    groups = h.mydet_dwconv_se_groups(Ho, Ho, C, K, 1)
    S = h.mydet_dwconv_slices(Ho, Ho, C, K, 1)
    assert groups > 0 and S > 0
    need = 4 * (_lib.SE_EPOCH_WORDS + 2 * B * groups * Cse)
    fake = 0x10000                                          # aligned, non-null, never touched
    for nbytes, want in ((need - 8, -1), (0, -1)):
        t = _lib.SeTail(fake, fake, fake, fake, fake, fake, Cse, nbytes)
        code = h.mydet_dwconv_f32(fake, C, fake, fake, fake, fake, C, B, Ho, Ho, C, K, 1, 2, 2, Ho, Ho, 2, None, S, ctypes.byref(t), None)
        assert code == want, (nbytes, code)
    t = _lib.SeTail(fake, fake, fake, fake, fake, fake, 97, need * 4)         # Cse beyond the kernels' limit: unsupported, not a launch
    assert h.mydet_dwconv_f32(fake, C, fake, fake, fake, fake, C, B, Ho, Ho, C, K, 1, 2, 2, Ho, Ho, 2, None, S, ctypes.byref(t), None) == -2


def test_no_store_data_hazard_in_the_shipped_library():
    """gfx950 store-data hazard (profiles/r03_isa_notes.md): no 12/16-byte buffer store with an SGPR soffset may be followed
    directly by a VALU write of its data registers -- hipcc pads that pattern only for stores WITHOUT an SGPR soffset.
    Checked on the disassembly of every gfx950 code object in libmydet_hip.so (tools/check_store_hazard.py)."""
    import importlib.util
    from mydetection_amd import _lib
    if not os.path.exists('/opt/rocm/lib/llvm/bin/llvm-objdump'):
        pytest.skip('no llvm-objdump on this machine')
    spec = importlib.util.spec_from_file_location('check_store_hazard', os.path.join(ROOT, 'tools', 'check_store_hazard.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    stores = hazards = objects = 0
    for _, text in mod.code_objects(_lib.LIB_PATH):
        a, _, bad = mod.violations(text)
        objects += 1
        stores += a
        hazards += len(bad)
    assert objects >= 10 and stores > 100, (objects, stores)
    assert hazards == 0


def test_no_packed_f32_valu_in_the_shipped_library():
    """No v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 in any kernel of libmydet_hip.so (csrc/Makefile: NOPK for every translation
    unit).  On MI355X the packed-f32 fma of one wave returned wrong low halves in lanes 48-63 while another wave of the SIMD
    issued dense bf16 MFMAs (the other batch lane's split-bf16 convs; profiles/r06_pk_fma_finding.md, tools/hw_pk_probe.hip):
    every kernel here can share a CU with those convs, so none may contain the packed forms -- no allow-list."""
    import importlib.util
    from mydetection_amd import _lib
    if not os.path.exists('/opt/rocm/lib/llvm/bin/llvm-objdump'):
        pytest.skip('no llvm-objdump on this machine')
    spec = importlib.util.spec_from_file_location('check_store_hazard', os.path.join(ROOT, 'tools', 'check_store_hazard.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.PACKED_F32.match('v_pk_fma_f32 v[0:1], v[10:11], v[48:49], v[0:1] op_sel_hi:[1,0,1]')
    assert not mod.PACKED_F32.match('v_pk_mov_b32 v[0:1], v[2:3], v[4:5]')
    users = mod.packed_f32_users(_lib.LIB_PATH)
    assert users == {}, sorted(users.items(), key=lambda kv: -kv[1])[:10]


def test_no_kernel_spills_beyond_a_few_registers():
    """Scratch (private segment) use of every kernel in the shipped library, read from the code objects' metadata
    (tools/check_store_hazard.py: scratch_users).  A spilled register set runs through memory on every use: round 5 found the
    split-K fixup kernels at 1-3 KB per lane (an array the unroller kept whole), one of them taking 89 us for a 32 MB sum.
    Allowed: the three 16-byte spills in the EPILOGUE of the F(4x4) GEMM kernel's residual variant (after the K loop; at the
    256-register limit since round 2) and in the opt-in wide split-bf16 form's fixups -- nothing else, and nothing above 64 B."""
    import importlib.util
    import re
    from mydetection_amd import _lib
    if not os.path.exists('/opt/rocm/lib/llvm/bin/llvm-readelf'):
        pytest.skip('no llvm-readelf on this machine')
    spec = importlib.util.spec_from_file_location('check_store_hazard', os.path.join(ROOT, 'tools', 'check_store_hazard.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    users = mod.scratch_users(_lib.LIB_PATH)
    allowed = re.compile(r'conv_wino4_kernelILi\dELb1ELb0E|conv_fixup_kernelILi128ELi256E')
    for name, nbytes in users.items():
        assert nbytes <= 64 and allowed.search(name), (name, nbytes)


def test_pyramid_node_address_audit():
    """Host-side enumeration of every global address the fused pyramid-node kernel forms (csrc/sepconv.hip:
    sp_stage_halo / sp_read_m and the epilogue store) over the whole node table of EfficientDet-D1 and D1-FCOS2-ATSS at
    batch 16 / 32, 640x640 -- all five levels (80^2 ... 5^2), the three fusion kinds, every tile and halo entry:
    each read must fall inside its input tensor (absent inputs of a single-input node are NEVER dereferenced), each
    store inside the output.  Round 2 lost a GPU box to an LDS-DMA variant of this kernel that read at
    0xffffd0ef7000 = -(low 32 bits of a heap pointer): a NULL input pointer of a single-input node had entered 32-bit
    offset arithmetic; this is the check that was missing."""
    import numpy as np
    TS, HS, C = 8, 10, 88

    def halo_coords(H, W):
        """(iy, ix) of all halo entries of all tiles of an H x W map, clamped as the kernel clamps them."""
        ty, tx = np.meshgrid(np.arange((H + TS - 1) // TS), np.arange((W + TS - 1) // TS), indexing='ij')
        hy, hx = np.meshgrid(np.arange(HS), np.arange(HS), indexing='ij')
        iy = (ty.reshape(-1, 1) * TS - 1 + hy.reshape(1, -1)).reshape(-1)
        ix = (tx.reshape(-1, 1) * TS - 1 + hx.reshape(1, -1)).reshape(-1)
        return np.clip(iy, 0, H - 1), np.clip(ix, 0, W - 1)

    def reads(mode, H, W, B, ld):
        """element offsets (first float of the first / last channel quad) read from an input of `mode` for an H x W node"""
        cy, cx = halo_coords(H, W)
        if mode == 0:
            h, w, py, px = H, W, cy, cx
        elif mode == 1:
            h, w, py, px = H // 2, W // 2, cy >> 1, cx >> 1
        else:
            h, w = H * 2, W * 2
            kh, kw = np.meshgrid(np.arange(3), np.arange(3), indexing='ij')
            py = np.clip(cy.reshape(-1, 1) * 2 - 1 + kh.reshape(1, -1), 0, h - 1).reshape(-1)
            px = np.clip(cx.reshape(-1, 1) * 2 - 1 + kw.reshape(1, -1), 0, w - 1).reshape(-1)
        b = np.array([0, B - 1]).reshape(-1, 1)
        base = ((b * h + py.reshape(1, -1)) * w + px.reshape(1, -1)) * ld
        return base.min(), base.max() + (C // 4 - 1) * 4 + 3, B * h * w * ld

    kinds = {'tower': (0,), 'top-down': (0, 1), 'bottom-up': (0, 0, 2), 'coarsest out': (0, 2)}
    checked = 0
    for B in (16, 32):
        for hw in (80, 40, 20, 10, 5):
            for name, modes in kinds.items():
                if 1 in modes and hw % 2:
                    continue                              # nearest-2x inputs need an even map (the C ABI rejects the rest)
                for m in modes:
                    lo, hi, numel = reads(m, hw, hw, B, C)
                    assert 0 <= lo and hi < numel, (B, hw, name, m, lo, hi, numel)
                    checked += 1
                for cout, ldy in ((88, 88), (720, 756), (36, 756), (4, 84)):
                    oy = np.arange(hw)
                    off_max = (((B - 1) * hw + oy.max()) * hw + oy.max()) * ldy + cout - 1
                    assert off_max < B * hw * hw * ldy
    assert checked == 2 * (5 * 1 + 4 * 2 + 5 * 3 + 5 * 2)
    # the per-channel shift / scale reads of sepconv_decode_kernel (SP_DEC_BLOCK): a float4 at min(16 nb + nsub, Cout - 4)
    # for every 16-channel block nb of the node and nsub = 0, 4, 8, 12.  A box node's vectors hold A*4 floats (36 for the
    # nine RetinaNet anchors: the last block is a quarter full); a class node's A * nba * 16 (whole blocks per anchor)
    for cout in (36, 4, 9 * 5 * 16, 9 * 6 * 16, 88):
        nb = (cout + 15) // 16
        offs = np.minimum(np.arange(nb).reshape(-1, 1) * 16 + np.array([0, 4, 8, 12]).reshape(1, -1), cout - 4)
        assert offs.min() >= 0 and offs.max() + 3 < cout, (cout, offs.max())
    # and the ABI refuses what the kernel could not address: a missing input pointer inside n_in, odd maps under up2x
    from mydetection_amd import _lib
    node = _lib.SepconvNode()
    node.n_in, node.H, node.W, node.Cout, node.ldy = 2, 8, 8, 88, 88
    assert _lib.lib().mydet_sepconv_nodes_f32(1, ctypes.cast(ctypes.pointer(node), ctypes.c_void_p), 1, 88, None) == -1


def test_pointwise_operand_packing():
    """ops.pack_pointwise / pack_pointwise_per_anchor (pure tensor code): the documented operand order of include/mydet.h --
    packed[nb][kq][lane][e] = W[16 nb + (lane & 15)][4 (4 kq + e) + (lane >> 4)], zeros beyond Cout and C; per anchor
    every anchor's n_cls rows start a fresh run of whole 16-channel blocks (what mydet_sepconv_decode_retina_f32 walks)."""
    from mydetection_amd import ops
    g = torch.Generator().manual_seed(3)
    for cout, C in ((88, 88), (36, 88), (720, 88), (20, 24)):
        w = torch.randn(cout, C, generator=g)
        p = ops.pack_pointwise(w)
        nb, kq = (cout + 15) // 16, (C // 4 + 3) // 4
        assert p.shape == (nb, kq, 64, 4) and p.is_contiguous()
        for b_, q_, lane, e in ((0, 0, 0, 0), (nb - 1, kq - 1, 63, 3), (nb // 2, kq // 2, 37, 2), (0, kq - 1, 16, 1)):
            row, k = 16 * b_ + (lane & 15), 4 * (4 * q_ + e) + (lane >> 4)
            want = float(w[row, k]) if row < cout and k < C else 0.0
            assert float(p[b_, q_, lane, e]) == want, (cout, C, b_, q_, lane, e)
        assert abs(float(p.double().abs().sum()) - float(w.double().abs().sum())) < 1e-9      # nothing lost, nothing duplicated
    A, n_cls, C = 9, 80, 88
    w, sh = torch.randn(A * n_cls, C, generator=g), torch.randn(A * n_cls, generator=g)
    p, s = ops.pack_pointwise_per_anchor(w, sh, A, n_cls)
    assert p.shape == (A * 5, 6, 64, 4) and s.shape == (A * 80,)
    A, n_cls = 3, 90                                                    # 90 classes: 96-row runs, 6 zero rows per anchor
    w, sh = torch.randn(A * n_cls, C, generator=g), torch.randn(A * n_cls, generator=g)
    p, s = ops.pack_pointwise_per_anchor(w, sh, A, n_cls)
    assert p.shape == (A * 6, 6, 64, 4) and s.shape == (A * 96,)
    for a in range(A):
        assert torch.equal(s[a * 96:a * 96 + 90], sh[a * 90:(a + 1) * 90]) and float(s[a * 96 + 90:(a + 1) * 96].abs().sum()) == 0
        blk = p[a * 6 + 5]                                              # the anchor's last block: rows 80..95, 10 real ones
        for lane in (0, 9, 10, 15, 31, 63):
            row = 80 + (lane & 15)
            want = float(w[a * 90 + row, 4 * (4 * 1 + 2) + (lane >> 4)]) if row < 90 else 0.0
            assert float(blk[1, lane, 2]) == want


def test_decode_node_abi_and_argument_checks():
    """mydet_sepconv_decode_node mirrors the C struct (size / offsets), and the entry point refuses what the kernel does not
    cover before any launch: class counts outside 65..96, channel counts other than 88, inconsistent Cout, null pointers."""
    from mydetection_amd import _lib
    assert ctypes.sizeof(_lib.SepconvNode) == 136 and ctypes.sizeof(_lib.SepconvDecodeNode) == 160
    assert _lib.SepconvDecodeNode.kind.offset == 136 and _lib.SepconvDecodeNode.anchors_wh.offset == 144
    lib = _lib.lib()
    arr = (_lib.SepconvDecodeNode * 1)()
    ptr = ctypes.cast(arr, ctypes.c_void_p)
    fake = ctypes.c_void_p(16)
    f = lib.mydet_sepconv_decode_retina_f32
    assert f(1, ptr, 1, 88, 9, 80, 64, 64, None, None, None, 100, None) == -1          # null outputs
    assert f(1, ptr, 1, 64, 9, 80, 64, 64, fake, fake, fake, 100, None) == -2           # C != 88
    assert f(1, ptr, 1, 88, 9, 20, 64, 64, fake, fake, fake, 100, None) == -2           # n_cls outside 65..96
    assert f(1, ptr, 1, 88, 13, 80, 64, 64, fake, fake, fake, 100, None) == -2          # more anchors than the table holds
    arr[0].node.n_in, arr[0].node.H, arr[0].node.W, arr[0].node.Cout, arr[0].kind = 1, 8, 8, 123, 0
    assert f(1, ptr, 1, 88, 9, 80, 64, 64, fake, fake, fake, 9 * 64, None) == -1        # Cout != A * 16 * ceil(n_cls / 16)
    arr[0].node.Cout = 9 * 80
    assert f(1, ptr, 1, 88, 9, 80, 64, 64, fake, fake, fake, 9 * 64, None) == -1        # null input / weights


def test_bench_parity_check_logic():
    """bench.py's `parity_check` (the gate every bench line carries) on the CPU: records that equal the oracle's post-process
    of the same candidates pass; a dropped detection, a swapped pair or a wrong class fails."""
    import importlib.util
    import numpy as np
    from oracle import postprocess as opp
    spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    rng = np.random.Generator(np.random.PCG64(3))
    B, N = 2, 3000
    bb = np.concatenate([rng.random((B, N, 2), dtype=np.float32) * 500 + 50, rng.random((B, N, 2), dtype=np.float32) * 80 + 10], axis=-1)
    ci = rng.integers(0, 20, size=(B, N)).astype(np.int64)
    sc = (rng.random((B, N), dtype=np.float32) ** 4).astype(np.float32)
    conf, nms = 0.05, 0.45
    count = np.zeros(B, dtype=np.int32)
    index = np.zeros((B, 512), dtype=np.int32)
    cls = np.zeros((B, 512), dtype=np.int64)
    for b in range(B):
        _, oc, _, oi = opp.post_process(bb[b], ci[b], sc[b], conf, nms)
        count[b], index[b, :len(oi)], cls[b, :len(oi)] = len(oi), oi, oc
    cand = tuple(torch.from_numpy(a) for a in (bb, ci, sc))
    rec = {'count': torch.from_numpy(count), 'index': torch.from_numpy(index), 'class_idx': torch.from_numpy(cls)}
    out = bench.parity_check(cand, rec, conf, nms, oracle_cand=None, images=B)
    assert out['ok'] and out['nms_equals_oracle_on_gpu_candidates'] and out['images'] == B
    assert count.min() >= 4
    for tamper in ('drop', 'swap', 'class'):
        r2 = {k: v.clone() for k, v in rec.items()}
        if tamper == 'drop':
            r2['count'][1] -= 1
        elif tamper == 'swap':
            r2['index'][0, [0, 1]] = r2['index'][0, [1, 0]]
        else:
            r2['class_idx'][1, 2] += 1
        assert not bench.parity_check(cand, r2, conf, nms, oracle_cand=None, images=B)['ok'], tamper


def test_bench_multi_gpu_launch_guard():
    """`python bench.py --gpus 8` on a machine without eight GPUs (this container has none): the launcher command is the
    driver's (`torch.distributed.run`, one rank per GPU, rendezvous on 127.0.0.1) and is built before anything touches a
    GPU; with fewer GPUs visible the process exits 2 with a one-line message and starts nothing."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    bench = os.path.join(root, 'bench.py')
    dry = subprocess.run([sys.executable, bench, '--gpus', '8', '--steps', '3', '--dry-run-launch'], capture_output=True, text=True,
                         timeout=120, env={k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')})
    assert dry.returncode == 0, dry.stderr
    cmd = json.loads(dry.stdout.strip().splitlines()[-1])['launch']
    assert cmd[1:4] == ['-m', 'torch.distributed.run', '--nnodes=1'] and '--nproc-per-node=8' in cmd
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1' and cmd[cmd.index('--master-port') + 1].isdigit()
    assert cmd[cmd.index(bench) + 1:] == ['--gpus', '8', '--steps', '3', '--dry-run-launch']
    if torch.cuda.device_count() >= 8:
        return                                     # an 8-GPU node would really start the ranks: nothing more to check here
    run = subprocess.run([sys.executable, bench, '--gpus', '8', '--steps', '3'], capture_output=True, text=True, timeout=120,
                         env={k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')})
    assert run.returncode == 2 and run.stdout.strip() == ''
    lines = [ln for ln in run.stderr.strip().splitlines() if ln.startswith('bench.py:')]
    assert len(lines) == 1 and '--gpus 8' in lines[0] and 'nothing was run' in lines[0]


def _w4_items(B, H, W, Cout):
    """(item validity by id, ids) of the F(4x4) GEMM launch, as conv_wino4.hip maps ids to (tile block, channel block)."""
    MT = B * ((H + 3) // 4) * ((W + 3) // 4)
    nmb, ntn = (MT + 31) // 32, (Cout + 31) // 32
    rn = 0
    while (1 << rn) < ntn and rn < 3:
        rn += 1
    RN, RM = 1 << rn, 64 >> rn
    nbn, nbm = (ntn + RN - 1) // RN, (nmb + RM - 1) // RM

    def valid(i):
        bi, w = i >> 6, i & 63
        mb, nb = (bi // nbn) * RM + (w >> rn), (bi % nbn) * RN + (w & (RN - 1))
        return mb < nmb and nb < ntn
    return valid, nbm * nbn * 64, nmb * ntn


def test_wino4_tail_plan_covers_every_item_once():
    """The K-cut tail plan of the F(4x4) GEMM launch (mydet_wino4_tail_plan: host only).  For a spread of shapes -- the three
    models' layers, ragged maps and channel counts, other chip sizes -- the main launch and the groups together take every
    valid item exactly once, the main launch is whole rounds of the chip (less at most a block), every group fits the chip once, pieces keep at
    least four K stages, and the partial-tile scratch areas are disjoint and inside the 64 MiB the workspace reserves."""
    from mydetection_amd import _lib
    lib = _lib.lib()
    shapes = [(32, 80, 80, 128, 256), (32, 40, 40, 256, 512), (32, 20, 20, 512, 1024), (32, 160, 160, 64, 128),
              (32, 64, 64, 128, 256), (32, 32, 32, 256, 512), (32, 16, 16, 512, 1024), (17, 32, 32, 64, 512),
              (24, 37, 37, 136, 200), (16, 20, 20, 88, 88), (32, 10, 10, 88, 88), (1, 32, 32, 256, 512)]
    rng = __import__('random').Random(4)
    for _ in range(300):
        shapes.append((rng.randint(1, 40), rng.randint(5, 90), rng.randint(5, 90), 4 * rng.randint(8, 160), 4 * rng.randint(2, 260)))
    seen_tail = 0
    for B, H, W, Cin, Cout in shapes:
        for slots in (512, 256, 608):
            out = (ctypes.c_int32 * 17)()
            ng = lib.mydet_wino4_tail_plan(B, H, W, Cin, Cout, slots, out)
            assert 0 <= ng <= 3, (B, H, W, Cin, Cout, ng)
            valid, nids, T = _w4_items(B, H, W, Cout)
            if ng == 0:
                assert out[0] == nids
                continue
            seen_tail += 1
            taken = [0] * nids
            for i in range(out[0]):
                taken[i] += 1
            main = sum(1 for i in range(out[0]) if valid(i))       # whole rounds, short of them by less than one block
            assert T // slots * slots - 64 < main <= T // slots * slots and out[0] % 64 == 0
            areas, nk = [], Cin // 4
            for g in range(ng):
                id0, blocks, stride, splits, off_kb = out[2 + 5 * g: 7 + 5 * g]
                assert id0 % 64 == 0 and 1 <= stride <= 64 and 2 <= splits <= 8 and nk // splits >= 4
                n = 0
                for b in range(blocks):
                    for w in range(64):
                        i = id0 + 64 * b + w
                        if w < stride:
                            taken[i] += 1
                            n += valid(i)
                        else:
                            assert not valid(i), 'a valid item lies beyond the ids the group takes from its block'
                assert 0 < n * splits <= slots
                areas.append((off_kb, off_kb + blocks * stride * splits * 64))          # 64 KiB per (slot, piece)
            assert all(t == 1 if valid(i) else t <= 1 for i, t in enumerate(taken)), (B, H, W, Cin, Cout, slots)
            areas.sort()
            assert areas[0][0] == 0 and areas[-1][1] <= 64 * 1024
            assert all(a[1] <= b[0] for a, b in zip(areas, areas[1:]))
    assert seen_tail >= 20                       # the rule does trigger on this set (the 40^2 / 20^2 layers at least)
    out = (ctypes.c_int32 * 17)()
    assert lib.mydet_wino4_tail_plan(32, 40, 40, 256, 512, 512, out) == 1 and list(out[2:6]) == [1536, 2, 32, 8]
    assert lib.mydet_wino4_tail_plan(32, 20, 20, 512, 1024, 512, out) == 2 and list(out[2:6]) == [512, 4, 64, 2] and list(out[7:11]) == [768, 4, 8, 8]
    assert lib.mydet_wino4_tail_plan(32, 80, 80, 128, 256, 512, out) == 0          # six whole rounds: no tail


def test_conv_p3_dispatch_rule():
    """ops.p3_takes: which 3x3 layers run on the patch-resident split-bf16 kernel (mydetection_amd/ops.py; measured in
    profiles/r06_conv_p3.txt): whole 8 x 16-pixel output tiles, a production-sized launch, stride 2 or a stride-1 layer below the
    F(4x4) channel limit."""
    from mydetection_amd import ops
    if not (ops.SPLIT_BF16 and ops.CONV_P3):
        return
    pad = (1, 1, 1, 1)
    # the headline (batch 32, 640^2): the first three stride-2 layers and the 32 -> 64 layer of the first DarkBlock ...
    for B, Ho, Cin, Cout, s in ((32, 320, 32, 64, 2), (32, 160, 64, 128, 2), (32, 80, 128, 256, 2), (32, 320, 32, 64, 1)):
        assert ops.p3_takes(B, Ho, Ho, Cin, Cout, 3, s, pad), (B, Ho, Cin, Cout, s)
    # ... and, with strip tiles for their remainder columns, the 40- and 20-pixel maps (13 / 4 tiles for 1 600 / 400 pixels)
    assert ops.p3_tiles(40, 40, 2) == 13 and ops.p3_tiles(20, 20, 2) == 4 and ops.p3_tiles(80, 80, 2) == 50 and ops.p3_tiles(20, 20, 1) == 6
    for B, Ho, Cin, Cout, s in ((32, 40, 256, 512, 2), (32, 20, 512, 1024, 2)):
        assert ops.p3_takes(B, Ho, Ho, Cin, Cout, 3, s, pad), (B, Ho, Cin, Cout, s)
    # not stride-1 layers F(4x4) takes, not small launches, not maps whose tiles would be mostly padding
    for B, Ho, Cin, Cout, s in ((32, 160, 64, 128, 1), (32, 80, 128, 256, 1), (1, 128, 64, 128, 2), (2, 64, 128, 256, 2), (64, 20, 32, 64, 1),
                                (400, 10, 64, 128, 2)):
        assert not ops.p3_takes(B, Ho, Ho, Cin, Cout, 3, s, pad), (B, Ho, Cin, Cout, s)
    # batch 1 at 512^2 (configs[0] shape): the first stride-2 layer and the 32 -> 64 layer fill a round of 512 workgroups
    assert ops.p3_takes(1, 256, 256, 32, 64, 3, 2, pad) and ops.p3_takes(1, 256, 256, 32, 64, 3, 1, pad)
    # only 3x3, pad 1, stride 1 | 2, Cin % 16 == 0
    assert not ops.p3_takes(32, 320, 320, 32, 64, 1, 1, (0, 0, 0, 0)) and not ops.p3_takes(32, 320, 320, 32, 64, 3, 2, (0, 0, 1, 1))
    assert not ops.p3_takes(32, 320, 320, 24, 64, 3, 2, pad) and not ops.p3_takes(32, 320, 320, 32, 64, 3, 3, pad)


def test_split_bf16_dispatch_rule():
    """ops.b3_takes: which direct-conv layers run on the split-bf16 kernel (mydetection_amd/ops.py: measured thresholds)."""
    from mydetection_amd import ops
    if not ops.SPLIT_BF16:
        return
    px = lambda b, s: b * s * s                                                       # noqa: E731
    # the headline (batch 32, 640^2): stride-2 3x3 layers, the DarkBlock / pyramid 1x1 layers from 128 outputs, the heads
    for M, cin, cout, k in ((px(32, 320), 32, 64, 3), (px(32, 20), 512, 1024, 3), (px(32, 80), 256, 128, 1), (px(32, 20), 1024, 512, 1),
                            (px(32, 20), 512, 256, 1), (px(32, 20), 1024, 255, 1), (px(32, 16), 512, 1024, 3)):
        assert ops.b3_takes(M, cin, cout, k), (M, cin, cout, k)
    # not: 1x1 layers below 128 outputs, channel counts that are no multiple of 16, and every layer of batch 1 at 512^2
    for M, cin, cout, k in ((px(32, 160), 128, 64, 1), (px(32, 320), 64, 32, 1), (px(32, 40), 40, 240, 1), (px(1, 256), 32, 64, 3),
                            (px(1, 128), 64, 128, 3), (px(1, 64), 128, 256, 3), (px(1, 64), 256, 128, 1)):
        assert not ops.b3_takes(M, cin, cout, k), (M, cin, cout, k)
    # the EfficientNet expand convs carry their own row limit: a lane of 8 images at 20^2 is in, a 10^2 map is not
    assert ops.b3_takes(px(8, 20), 192, 1152, 1, ops.B3_EXPAND_MIN_ROWS) and not ops.b3_takes(px(8, 10), 192, 1152, 1, ops.B3_EXPAND_MIN_ROWS)
    # the gated project convs: from 64 output channels and 3 000 rows (a lane of 8 images at 20^2), not the 40-channel layers
    assert ops.b3_takes(px(16, 40), 672, 112, 1, ops.B3_GATED_MIN_ROWS, min_cout=ops.B3_GATED_MIN_COUT)
    assert ops.b3_takes(px(8, 20), 1152, 192, 1, ops.B3_GATED_MIN_ROWS, min_cout=ops.B3_GATED_MIN_COUT)
    assert not ops.b3_takes(px(16, 80), 240, 40, 1, ops.B3_GATED_MIN_ROWS, min_cout=ops.B3_GATED_MIN_COUT)
