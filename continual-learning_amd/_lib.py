"""ctypes binding of libclamd.so (the C ABI declared in include/clamd.h).

There is NO fallback: if the library is missing or a call fails, a RuntimeError is raised.  The product path never
routes through torch operators or the CPU oracle for compute.
"""
import ctypes
import os
from ctypes import c_char_p, c_double, c_int, c_longlong, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# CLAMD_LIB: an experiment build of the same sources (build.py --variant), for the A/B and diagnostic tools only
LIB_PATH = os.environ.get('CLAMD_LIB') or os.path.join(_HERE, 'libclamd.so')

F32, BF16, SPLIT = 0, 1, 2      # SPLIT = 'bf16x3': fp32 storage, 3-term split-bf16 MFMA
WGRAD_CONV3, WGRAD_PW, WGRAD_UP2 = 0, 1, 2

_P, _I, _D, _LL, _SZ = c_void_p, c_int, c_double, c_longlong, c_size_t
OP_CONV3X3, OP_CONV3X3_WINOGRAD, OP_CONV1X1, OP_CONVT2X2_DGRAD, OP_BN_BWD_REDUCE, OP_CONV3X3_WINOGRAD24, OP_CONV3X3_WINOGRAD44 = range(7)


class Tuning(ctypes.Structure):
    """Mirror of `clamd_tuning` (include/clamd.h): per-call kernel-structure selection.  The library keeps no process
    state; an engine owns one of these and passes it to every launch (None = library defaults)."""
    _fields_ = [(n, c_int) for n in ('igemm_pws', 'igemm_ws', 'igemm_variant', 'pws_wres', 'wgrad_ws', 'wgrad_dma',
                                     'wgrad_xcd', 'wgrad_blocks', 'wgrad_tw16', 'wino_band', 'wino_persist', 'wino_mt',
                                     'bn_reduce_blocks', 'chsum_blocks', 'cu_reserve', 'wino_half', 'wgrad_streamk')] + [('reserved', c_int * 7)]

    def __init__(self, **kw):
        super().__init__()
        load().clamd_tuning_init(ctypes.byref(self))
        for k, v in kw.items():
            if k not in dict(self._fields_):
                raise KeyError(f'unknown tuning field {k!r}')
            setattr(self, k, int(v))

    def ref(self):
        return ctypes.addressof(self)

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_ if n != 'reserved'}


def tune_ptr(t):
    """`const clamd_tuning*` argument for a Tuning object (None -> NULL = library defaults)."""
    return None if t is None else t.ref()


# name -> (restype, argtypes); must list EVERY symbol of include/clamd.h (tests/test_host_cpu.py checks the header).
SIGNATURES = {
    'clamd_last_error': (c_char_p, []),
    'clamd_version': (_I, []),
    'clamd_sizeof_pack_job': (_I, []),
    'clamd_sizeof_adam_tensor': (_I, []),
    'clamd_adam_chunk_elems': (_I, []),
    'clamd_pack_tile': (_I, []),
    'clamd_bn_bwd_nsums': (_I, []),
    'clamd_sizeof_tuning': (_I, []),
    'clamd_tuning_init': (None, [_P]),
    'clamd_stat_rows': (_I, [_I, _I, _I, _I, _I, _I, _I, _I, _P]),
    'clamd_conv3x3': (_I, [_P, _I, _P, _P, _P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P]),
    'clamd_conv3x3_border_bias_ok': (_I, [_I, _I, _I, _I, _I, _I, _P]),
    'clamd_conv3x3_bn_sums': (_I, [_I, _I, _I, _I, _I, _I, _P]),
    'clamd_bn_fold_bias': (_I, [_P, _P, _P, _P, _I, _I, _I, _P]),
    'clamd_maxpool2x2': (_I, [_P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    'clamd_maxpool2x2_bwd': (_I, [_P, _I, _P, _P, _I, _P, _I, _I, _I, _I, _I, _I, _P]),
    'clamd_bn_fold_pack': (_I, [_I, _P, _I, _I, _I, _P, _I, _P, _P, _P, _I, _I, _I, _P]),
    'clamd_bn_fold_wgrad_pointwise': (_I, [_P, _P, _P, _P, _I, _I, _P]),
    'clamd_bn_fold_wgrad_workspace_bytes': (_SZ, [_I, _I]),
    'clamd_bn_fold_wgrad': (_I, [_P, _I, _P, _P, _P, _P, _P, _SZ, _I, _I, _I, _I, _I, _I, _I, _P]),
    'clamd_conv1x1': (_I, [_P, _I, _P, _P, _P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    'clamd_conv1x1_logits': (_I, [_P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    'clamd_conv1x1_argmax': (_I, [_P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    'clamd_convT2x2_fwd': (_I, [_P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    'clamd_convT2x2_dgrad': (_I, [_P, _I, _P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    'clamd_wgrad_workspace_bytes': (_SZ, [_I, _I, _I, _I, _I, _I, _I]),
    'clamd_wgrad': (_I, [_I, _P, _I, _P, _I, _P, _SZ, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P]),
    'clamd_bn_finalize': (_I, [_P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _D, _D, _D, _P, _P]),
    'clamd_bn_apply': (_I, [_P, _I, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _I, _P]),
    'clamd_bn_bwd_reduce': (_I, [_P, _I, _P, _I, _P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P]),
    'clamd_bn_bwd_finalize': (_I, [_P, _I, _P, _P, _P, _P, _P, _P, _P, _I, _I, _D, _P]),
    'clamd_bn_bwd_apply': (_I, [_P, _I, _P, _I, _P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    'clamd_bn_bwd_apply_sums_rows': (_I, [_I, _I, _I, _I]),
    'clamd_bn_bwd_apply_sums': (_I, [_P, _I, _P, _I, _P, _P, _I, _P, _I, _I, _I, _I, _I, _I, _P]),
    'clamd_rows_sum': (_I, [_P, _I, _P, _I, _I, _P]),
    'clamd_channel_sum_workspace_bytes': (_SZ, [_I]),
    'clamd_channel_sum': (_I, [_P, _I, _P, _LL, _I, _I, _I, _P, _SZ, _P, _P]),
    'clamd_nchw_to_nhwc': (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _D, _I, _P]),
    'clamd_nhwc_to_nchw': (_I, [_P, _I, _P, _I, _I, _I, _I, _I, _P]),
    'clamd_nchw_im2col3': (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    'clamd_pack': (_I, [_P, _I, _I, _I, _P]),
    'clamd_sizeof_wino_pack_job': (_I, []),
    'clamd_wino_pack': (_I, [_P, _I, _I, _P]),
    'clamd_wgrad_winograd_workspace_bytes': (_SZ, [_I, _I]),
    'clamd_wgrad_winograd24_workspace_bytes': (_SZ, [_I, _I]),
    'clamd_wgrad_winograd24': (_I, [_P, _I, _P, _I, _P, _SZ, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P]),
    'clamd_wgrad_winograd': (_I, [_P, _I, _P, _I, _P, _SZ, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P]),
    'clamd_conv3x3_winograd': (_I, [_P, _I, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P]),
    'clamd_wino24_pack': (_I, [_P, _I, _I, _P]),
    'clamd_conv3x3_winograd24': (_I, [_P, _I, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P]),
    'clamd_winograd24_input_elems': (_SZ, [_I, _I, _I, _I]),
    'clamd_winograd24_transform_input': (_I, [_P, _I, _P, _P, _P, _I, _I, _I, _I, _P]),
    'clamd_conv3x3_winograd24_pre': (_I, [_P, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P]),
    'clamd_conv3x3_winograd24_direct_filters': (_I, [_P, _I, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P]),
    'clamd_wgrad_winograd24_pre_operand_elems': (_SZ, [_I, _I, _I, _I]),
    'clamd_wgrad_winograd24_pre_transform': (_I, [_P, _I, _P, _I, _I, _I, _I, _P]),
    'clamd_wgrad_winograd24_pre_workspace_bytes': (_SZ, [_I, _I, _I, _I, _I]),
    'clamd_wgrad_winograd24_pre': (_I, [_P, _I, _P, _P, _P, _SZ, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P]),
    'clamd_wino44_pack': (_I, [_P, _I, _I, _P]),
    'clamd_winograd44_input_elems': (_SZ, [_I, _I, _I, _I]),
    'clamd_winograd44_transform_input': (_I, [_P, _I, _P, _P, _P, _I, _I, _I, _I, _P]),
    'clamd_conv3x3_winograd44_pre': (_I, [_P, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P]),
    'clamd_wgrad_winograd44_pre_operand_elems': (_SZ, [_I, _I, _I, _I]),
    'clamd_wgrad_winograd44_pre_transform': (_I, [_P, _I, _P, _I, _I, _I, _I, _P]),
    'clamd_wgrad_winograd44_pre_workspace_bytes': (_SZ, [_I, _I, _I, _I, _I]),
    'clamd_wgrad_winograd44_pre': (_I, [_P, _I, _P, _P, _P, _SZ, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P]),
    'clamd_ce_workspace_bytes': (_SZ, []),
    'clamd_ce_bad_label_count_offset': (_SZ, []),
    'clamd_ce_fwd_bwd': (_I, [_P, _P, _P, _I, _I, _D, _D, _P, _P, _P, _SZ, _I, _I, _I, _I, _LL, _D, _P]),
    'clamd_ce_count': (_I, [_P, _I, _I, _I, _I, _LL, _P, _SZ, _P]),
    'clamd_ce_fwd_bwd_counted': (_I, [_P, _P, _P, _P, _I, _I, _P, _P, _SZ, _I, _I, _I, _I, _LL, _D, _P]),
    'clamd_adam_step': (_I, [_P, _P, _I, _P, _P, _P, _P, _P]),
    'clamd_argmax_confusion': (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    'clamd_voc_prepare': (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P]),
    'clamd_label_to_rgb': (_I, [_P, _P, _LL, _LL, _P]),
    'clamd_fill_f32': (_I, [_P, _LL, _D, _P]),
    'clamd_hold_cus': (_I, [_I, _I, _P]),
    'clamd_f32_to_bf16': (_I, [_P, _P, _LL, _P]),
    'clamd_bf16_to_f32': (_I, [_P, _P, _LL, _P]),
    'clamd_scale_by_device_scalar': (_I, [_P, _LL, _P, _P]),
    'clamd_scale_by_device_scalar_nhwc': (_I, [_P, _LL, _I, _P, _P, _LL, _P]),
}

# include/clamd_debug.h: measurement scaffolding (tools/cu_steal.py), bound when present, never part of the product header
DEBUG_SIGNATURES = {'clamd_debug_hold_cus': (_I, [_I, _I, _P]), 'clamd_debug_mfma_rate': (_LL, [_I, _I, _P, _P])}

_lib = None


def load():
    """Loads the library (once).  Raises RuntimeError if it has not been built (run __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f'{LIB_PATH} is missing: the HIP extension has not been built '
                           f'(python continual-learning_amd/build.py).  There is no CPU/torch fallback.')
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    for name, (res, args) in DEBUG_SIGNATURES.items():
        fn = getattr(lib, name, None)
        if fn is not None:
            fn.restype = res
            fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=''):
    if rc != 0:
        msg = load().clamd_last_error().decode()
        raise RuntimeError(f'libclamd {what} failed ({rc}): {msg}')


def call(name, *args):
    """Invoke a status-returning entry point and raise on error."""
    check(getattr(load(), name)(*args), name)


def stat_rows(op, B, H, W, cin_p, cout_p, dcode, fused_bn=False, tuning=None):
    """Partial-statistics rows a launch of `op` writes (clamd_stat_rows); raises on error."""
    n = load().clamd_stat_rows(op, B, H, W, cin_p, cout_p, dcode, 1 if fused_bn else 0, tune_ptr(tuning))
    if n <= 0:
        check(n, 'clamd_stat_rows')
    return n


def ptr(t):
    """Raw device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def stream_ptr():
    import torch
    return torch.cuda.current_stream().cuda_stream
