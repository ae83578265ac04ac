"""ctypes binding of libmydet_hip.so (C ABI declared in include/mydet.h).

The library is the product: there is no CPU or PyTorch fallback.  If it is missing
the import of any op fails loudly with instructions to build it.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MYDET_LIB_PATH: a differently built copy of the library (kernel experiments: tools/ablate_sepconv.sh); default = the in-tree build
LIB_PATH = os.environ.get('MYDET_LIB_PATH') or os.path.join(_HERE, 'lib', 'libmydet_hip.so')

c_int, c_i64, c_f32, c_f64, c_ptr = ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_double, ctypes.c_void_p

# name -> argtypes, exactly the prototypes of include/mydet.h
SIGNATURES = {
    'mydet_abi_version': [],
    'mydet_conv2d_igemm_f32': [c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_ptr, c_i64, c_ptr, c_i64]
    + [c_int] * 13 + [c_ptr],
    'mydet_wino_weights_floats': [c_int, c_int],
    'mydet_wino_weights_f32': [c_ptr, c_int, c_int, c_ptr, c_ptr],
    'mydet_conv2d_wino_f32': [c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64] + [c_int] * 6
    + [c_ptr],
    'mydet_wino4_weights_floats': [c_int, c_int],
    'mydet_wino4_weights_f32': [c_ptr, c_int, c_int, c_ptr, c_ptr],
    'mydet_wino4_workspace_bytes': [c_int, c_int, c_int, c_int, c_int],
    'mydet_wino4_reload_tuning': [],
    'mydet_conv_b3_reload_tuning': [],
    'mydet_conv_igemm_occupancy': [c_int, c_ptr],
    'mydet_split_bf16_elems': [c_int, c_int],
    'mydet_split_bf16_f32': [c_ptr, c_int, c_int, c_ptr, c_ptr],
    'mydet_conv2d_igemm_b3_f32': [c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_ptr, c_i64, c_ptr, c_i64] + [c_int] * 13 + [c_ptr],
    'mydet_conv3x3_p3_f32': [c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_i64] + [c_int] * 7 + [c_ptr],
    'mydet_wino4_tail_plan': [c_int, c_int, c_int, c_int, c_int, c_int, c_ptr],
    'mydet_conv2d_wino4_f32': [c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64] + [c_int] * 6 + [c_ptr],
    'mydet_dwconv_slices': [c_int, c_int, c_int, c_int, c_int],
    'mydet_dwconv_se_groups': [c_int, c_int, c_int, c_int, c_int],
    'mydet_dwconv_f32': [c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_i64] + [c_int] * 11 + [c_ptr, c_int, c_ptr, c_ptr],
    'mydet_channel_sums_f32': [c_ptr, c_i64, c_int, c_int, c_int, c_int, c_ptr, c_int, c_ptr],
    'mydet_se_gate_f32': [c_ptr, c_int, c_int, c_int, c_int, c_ptr, c_ptr, c_int, c_ptr, c_ptr, c_ptr, c_ptr],
    'mydet_maxpool3s2_f32': [c_ptr, c_i64, c_ptr, c_i64] + [c_int] * 6 + [c_ptr],
    'mydet_bifpn_fuse_f32': [c_int, c_ptr, c_i64, c_int, c_ptr, c_i64, c_int, c_ptr, c_i64, c_int, c_ptr, c_ptr, c_i64,
                             c_int, c_int, c_int, c_int, c_ptr],
    'mydet_conv2d_stem_f32': [c_ptr, c_i64, c_i64, c_i64, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_i64] + [c_int] * 10 + [c_ptr],
    'mydet_upsample_concat_f32': [c_ptr, c_i64, c_int, c_int, c_int, c_ptr, c_i64, c_int, c_ptr, c_i64, c_int, c_int,
                                  c_int, c_ptr],
    'mydet_conv1x1_upcat_f32': [c_ptr, c_i64, c_int, c_ptr, c_i64, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_i64]
    + [c_int] * 5 + [c_ptr],
    'mydet_space_to_depth_f32': [c_ptr, c_i64, c_i64, c_i64, c_i64, c_ptr, c_i64, c_int, c_int, c_int, c_int, c_ptr],
    'mydet_spp_concat_f32': [c_ptr, c_i64, c_ptr, c_i64] + [c_int] * 7 + [c_ptr],
    'mydet_decode_levels_f32': [c_int, c_int, c_ptr, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                c_ptr, c_ptr, c_ptr, c_i64, c_ptr],
    'mydet_decode_f32': [c_int, c_ptr, c_i64, c_int, c_int, c_ptr, c_i64, c_int, c_int, c_int, c_ptr, c_int, c_int,
                         c_int, c_int, c_int, c_f32, c_int, c_int, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_ptr],
    'mydet_postprocess_f32': [c_ptr, c_ptr, c_ptr, c_int, c_i64, c_f32, c_f64, c_int, c_ptr, c_ptr, c_ptr, c_ptr,
                              c_ptr, c_ptr, c_ptr],
    'mydet_postprocess_records_f32': [c_ptr, c_ptr, c_ptr, c_int, c_i64, c_f32, c_f64, c_ptr, c_ptr, c_ptr],
    'mydet_mbconv_tiles': [c_int, c_int, c_int],
    'mydet_mbconv_expand_dw_f32': [c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64] + [c_int] * 11 + [c_ptr, c_int, c_ptr, c_ptr],
    'mydet_sepconv_nodes_f32': [c_int, c_ptr, c_int, c_int, c_ptr],
    'mydet_stem_dw_f32': [c_ptr, c_i64, c_i64, c_i64, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_int, c_int, c_int, c_int,
                          c_int, c_int, c_int, c_int, c_ptr, c_int, c_ptr, c_ptr],
    'mydet_sepconv_decode_retina_f32': [c_int, c_ptr, c_int, c_int, c_int, c_int, c_int, c_int, c_ptr, c_ptr, c_ptr, c_i64, c_ptr],
    'mydet_bboxes_iou_f32': [c_ptr, c_int, c_ptr, c_int, c_int, c_ptr, c_ptr],
    'mydet_bboxes_to_original_batched_f32': [c_ptr, c_i64, c_ptr, c_i64, c_int, c_int, c_ptr, c_ptr],
    'mydet_detections_to_json_f64': [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_int, c_int, c_ptr, c_int, c_ptr, c_ptr, c_ptr],
    'mydet_resize_bilinear_u8': [c_ptr, c_int, c_int, c_i64, c_ptr, c_int, c_int, c_i64, c_ptr, c_ptr, c_int, c_ptr, c_ptr, c_int, c_ptr],
    'mydet_preprocess_u8_f32': [c_ptr, c_int, c_int, c_int, c_ptr, c_int, c_int, c_int, c_ptr, c_ptr, c_ptr],
    'mydet_cxcywh_to_x1y1x2y2_f32': [c_ptr, c_ptr, c_i64, c_int, c_ptr],
    'mydet_bboxes_to_original_f32': [c_ptr, c_i64, c_f32, c_f32, c_f32, c_f32, c_f32, c_f32, c_ptr],
}



RETURNS_I64 = {'mydet_wino_weights_floats', 'mydet_wino4_weights_floats', 'mydet_wino4_workspace_bytes', 'mydet_split_bf16_elems'}

# detection record layout (MYDET_REC_* of include/mydet.h), in int32 words
REC_TOPK = 512
REC_COUNT, REC_BBOX = 0, 4
REC_SCORE = REC_BBOX + 4 * REC_TOPK
REC_CLASS = REC_SCORE + REC_TOPK
REC_INDEX = REC_CLASS + 2 * REC_TOPK
REC_WORDS = REC_INDEX + REC_TOPK


class DecodeLevel(ctypes.Structure):
    """mydet_decode_level (include/mydet.h)."""
    _fields_ = [('box', c_ptr), ('ldbox', c_i64), ('cls', c_ptr), ('ldcls', c_i64), ('anchors_wh', c_ptr),
                ('H', c_int), ('W', c_int), ('stride', c_f32), ('n_off', c_i64)]


class SeTail(ctypes.Structure):
    """mydet_se_tail (include/mydet.h): the squeeze-excite tail finished inside the depthwise launch."""
    _fields_ = [('w1', c_ptr), ('b1', c_ptr), ('w2t', c_ptr), ('b2', c_ptr), ('gate', c_ptr), ('hpart', c_ptr), ('Cse', c_int),
                ('hpart_bytes', c_i64)]


SE_EPOCH_WORDS = 1024               # MYDET_SE_EPOCH_WORDS of include/mydet.h
ABI_VERSION = 2                     # MYDET_ABI_VERSION of include/mydet.h (2: mydet_se_tail.hpart_bytes, mydet_conv3x3_p3_f32)


class SepconvNode(ctypes.Structure):
    """mydet_sepconv_node (include/mydet.h)."""
    _fields_ = [('inp', c_ptr * 3), ('ld', c_i64 * 3), ('mode', c_int * 3), ('n_in', c_int), ('fuse_weights', c_ptr),
                ('w_dw', c_ptr), ('w_pw_packed', c_ptr), ('scale', c_ptr), ('shift', c_ptr), ('y', c_ptr), ('ldy', c_i64),
                ('H', c_int), ('W', c_int), ('Cout', c_int), ('act', c_int)]


class SepconvDecodeNode(ctypes.Structure):
    """mydet_sepconv_decode_node (include/mydet.h)."""
    _fields_ = [('node', SepconvNode), ('kind', c_int), ('stride', c_f32), ('anchors_wh', c_ptr), ('n_off', c_i64)]


SEPCONV_MAX_NODES = 10

_lib = None


class MissingHipLibrary(ImportError):
    pass


def lib():
    """Load (once) and return the bound library; raise MissingHipLibrary if it was not built."""
    global _lib
    if _lib is None:
        # PyTorch-ROCm ships its own libamdhip64; it must be in the process BEFORE this library is loaded so that
        # both resolve to ONE HIP runtime (otherwise launches here hit a runtime with no device: hipError 100).
        import torch  # noqa: F401
        if not os.path.exists(LIB_PATH):
            raise MissingHipLibrary(
                f'{LIB_PATH} not found. mydetection_amd has no CPU/PyTorch fallback: build the gfx950 '
                'kernels first with  python -c "import __graft_entry__ as g; g.build()"  '
                '(or make -C mydetection_amd/csrc).')
        handle = ctypes.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(handle, name)          # AttributeError here = ABI mismatch, also loud
            fn.argtypes = argtypes
            fn.restype = c_i64 if name in RETURNS_I64 else c_int
        if handle.mydet_abi_version() != ABI_VERSION:
            raise MissingHipLibrary(f'libmydet_hip.so has ABI version {handle.mydet_abi_version()}, this package binds version {ABI_VERSION}; rebuild it '
                                    '(make -C mydetection_amd/csrc)')
        _lib = handle
    return _lib


class MydetError(RuntimeError):
    pass


def check(code, what):
    if code != 0:
        kind = {-1: 'bad argument', -2: 'unsupported configuration'}.get(code, f'hipError {code}')
        raise MydetError(f'{what} failed: {kind}')
