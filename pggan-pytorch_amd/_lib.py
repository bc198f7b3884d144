"""ctypes binding of libpggan_hip.so (C-ABI declared in include/pggan_hip.h).

There is deliberately NO fallback: if the shared library is missing or a symbol is absent the
import of the product path fails loudly (``PgganLibraryError``).  Build it with
``python __graft_entry__.py`` (or ``__graft_entry__.build()``)."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('PGGAN_HIP_LIB') or os.path.join(_HERE, 'libpggan_hip.so')   # env override: kernel A/B experiments
ABI_VERSION = 25


class PgganLibraryError(RuntimeError):
    pass


P = ctypes.c_void_p
I = ctypes.c_int
L = ctypes.c_int64
F = ctypes.c_float
D = ctypes.c_double

# name -> argtypes (stream is always the last void*).  Mirrors include/pggan_hip.h 1:1; the thread-local diagnostic
# exports of include/pggan_hip_debug.h are listed in DEBUG_SIGNATURES.
SIGNATURES = {
    'pg_abi_version': [],
    'pg_conv2d_nhwc': [P, P, P, P, P, I, I, I, I, I, I, I, I, F, F, F, P],
    'pg_conv2d_pool_nhwc': [P, P, P, P, P, P, P, F, F, I, I, I, I, I, I, I, I, I, F, F, F, P],
    'pg_conv2d_pixelnorm_nhwc': [P, P, P, P, P, I, I, I, I, I, I, I, I, F, F, F, P],
    'pg_conv2d_pnbwd_nhwc': [P, P, P, P, P, I, I, I, I, I, I, I, F, F, P],
    'pg_conv2d_unpool_nhwc': [P, P, P, P, P, I, I, I, I, I, I, I, I, F, F, F, P],
    'pg_signbytes_to_mask': [P, P, L, P],
    'pg_conv2d_unpooled_nhwc': [P, P, P, F, F, P, P, I, I, I, I, I, I, F, F, P],
    'pg_conv2d_wgrad_unpooled_nhwc': [P, P, P, F, F, P, P, I, I, I, I, I, F, P],
    'pg_conv2d_pixelnorm_torgb_nhwc': [P, P, P, P, P, P, P, F, P, I, I, I, I, I, I, F, F, F, P],
    'pg_conv2d_masked_fromrgb_bwd_nhwc': [P, P, P, F, P, P, F, P, P, P, P, I, I, I, I, I, I, F, P],
    'pg_conv2d_fromrgb_nhwc': [P, P, P, F, F, P, P, P, P, P, I, I, I, I, I, I, F, F, P],
    'pg_wino_transform_weights': [P, P, I, I, P],
    'pg_wino_transform_weights_batched': [P, P, I, P, P, P, P, P, P],
    'pg_conv2d_wino_nhwc': [P, P, P, P, P, P, P, F, F, I, P, P, F, I, I, I, I, I, I, F, F, F, P],
    'pg_conv2d_wino_pixelnorm_nhwc': [P, P, P, P, P, I, I, I, I, I, I, F, F, F, P],
    'pg_conv2d_wino_pnbwd_nhwc': [P, P, P, P, P, I, P, F, F, I, I, I, I, I, F, F, P],
    'pg_set_workspace': [P, P, ctypes.c_size_t],
    'pg_workspace_bytes': [I, I, I, I, I, I, P],
    'pg_conv2d_wgrad_wino_nhwc': [P, P, P, P, I, I, I, I, I, I, F, P],
    'pg_conv2d_wgrad_wino2_nhwc': [P, P, I, P, P, I, P, P, I, I, I, I, I, I, F, P],
    'pg_conv2d_wgrad_nhwc': [P, P, P, P, I, I, I, I, I, I, I, I, F, P],
    'pg_pack_dgrad_weights': [P, P, I, I, I, P],
    'pg_pack_dgrad_weights_batched': [P, P, I, P, P, P, P, P],
    'pg_fromrgb_fwd': [P, P, P, P, P, I, I, I, I, I, I, F, F, F, P],
    'pg_fromrgb_bwd_data': [P, P, P, I, I, I, I, I, I, I, F, P],
    'pg_fromrgb_wgrad': [P, P, P, P, I, I, I, I, I, I, F, P],
    'pg_torgb_fwd': [P, P, P, P, P, I, I, I, I, I, F, F, F, P],
    'pg_torgb_bwd_data': [P, P, P, I, I, I, I, I, I, F, P],
    'pg_torgb_bwd_data_pnbwd': [P, P, P, P, P, I, I, I, I, I, F, F, P],
    'pg_torgb_wgrad': [P, P, P, P, I, I, I, I, I, I, F, F, P],
    'pg_avgpool2_fwd': [P, P, P, I, I, I, I, F, F, P],
    'pg_avgpool2_bwd': [P, P, P, I, I, I, I, F, F, P],
    'pg_upsample2_bwd': [P, P, I, I, I, I, P],
    'pg_axpby_mask': [P, P, P, P, L, F, F, F, P],
    'pg_pixelnorm_fwd': [P, P, P, L, I, F, P],
    'pg_pixelnorm_lrelu_bwd': [P, P, P, P, L, I, F, P],
    'pg_pixelnorm_tangent': [P, P, P, P, P, P, L, I, P],
    'pg_pixelnorm_lrelu_bwd_inj': [P, P, P, P, P, L, I, F, P],
    'pg_mbstd_fwd': [P, P, P, I, I, I, I, I, P],
    'pg_mbstd_tangent': [P, P, P, P, P, I, I, I, I, I, P],
    'pg_mbstd_bwd': [P, P, P, P, P, P, P, I, I, I, I, I, I, F, P],
    'pg_mbstd_stats': [P, P, I, I, I, I, P],
    'pg_mbstd_write': [P, P, P, P, I, I, I, I, I, I, P],
    'pg_mbstd_tangent_stats': [P, P, P, P, I, I, I, I, P],
    'pg_mbstd_tangent_write': [P, P, P, P, P, I, I, I, I, I, I, P],
    'pg_mbstd_gsum': [P, P, P, I, I, I, I, I, P],
    'pg_mbstd_bwd_global': [P, P, P, P, P, P, P, P, I, I, I, I, I, I, I, F, P],
    'pg_linear1_fwd': [P, P, P, P, I, I, P],
    'pg_linear1_bwd_data': [P, P, P, P, I, I, F, P],
    'pg_linear1_wgrad': [P, P, P, P, I, I, P],
    'pg_gp_mix': [P, P, P, P, I, L, P],
    'pg_row_sumsq': [P, P, I, L, P],
    'pg_gp_seed': [P, P, P, P, I, L, F, F, F, P],
    'pg_d_loss': [P, P, P, P, P, P, I, F, P],
    'pg_g_loss': [P, P, P, I, P],
    'pg_adam': [P, P, P, P, L, F, F, F, F, F, F, F, P],
    'pg_real_prepare_u8': [P, P, L, I, I, D, D, D, D, D, P],
    'pg_image_grid_u8': [P, P, I, I, I, I, I, F, F, P],
    'pg_pyramid_level_u8': [P, P, L, I, I, I, F, F, P],
    'pg_zero': [P, L, P],
    'pg_uniform_f32': [P, L, ctypes.c_uint64, ctypes.c_uint64, P],
    'pg_stft_abslog': [P, L, I, P, I, I, I, I, P],
    'pg_stft_image': [P, L, I, P, I, I, I, I, I, P],
    'pg_mono_f32': [P, L, I, P, L, P],
    'pg_minmax_f32': [P, L, P, P],
    'pg_stretch_to_u8': [P, P, L, P, F, P],
    # gradient exchange (RCCL bound at run time inside the library)
    'pg_rccl_version': [P],
    'pg_comm_unique_id': [P],
    'pg_comm_init_rank': [P, I, P, I],
    'pg_comm_info': [P, P, P],
    'pg_comm_destroy': [P],
    'pg_allreduce_sum_f32': [P, P, L, P],
}

DEBUG_SIGNATURES = {
    'pg_debug_last_conv_kernel': [],
    'pg_debug_last_wino_kernel': [],
    'pg_debug_last_wino_wgrad_kernel': [],
    'pg_debug_set_tuning': [I, I],
    'pg_debug_set_wino': [I],
    'pg_debug_set_wino_ksplit': [I],
    'pg_debug_set_wino_epi': [I],
}

_lib = None


def load():
    """Load the shared library once and declare every prototype.  Raises PgganLibraryError."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PgganLibraryError(
            'libpggan_hip.so not found at %s: the HIP extension is not built (run '
            '`python __graft_entry__.py`). There is no CPU fallback.' % LIB_PATH)
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:
        raise PgganLibraryError('cannot load %s: %s' % (LIB_PATH, e))
    for name, argtypes in list(SIGNATURES.items()) + list(DEBUG_SIGNATURES.items()):
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise PgganLibraryError('symbol %s missing from %s' % (name, LIB_PATH))
        fn.argtypes = argtypes
        fn.restype = ctypes.c_char_p if name in ('pg_debug_last_conv_kernel', 'pg_debug_last_wino_kernel', 'pg_debug_last_wino_wgrad_kernel') else ctypes.c_int
    v = lib.pg_abi_version()
    if v != ABI_VERSION:
        raise PgganLibraryError('ABI version mismatch: library %d, binding %d' % (v, ABI_VERSION))
    _lib = lib
    return lib


_ERR = {-1: 'PG_E_ARG (bad dimension / null pointer)', -2: 'PG_E_ALIGN (channel count / alignment)',
        -3: 'PG_E_UNSUP (unsupported configuration)', -4: 'PG_E_NOLIB (no RCCL library could be loaded)'}


class Unsupported(RuntimeError):
    """PG_E_UNSUP: the entry point does not take this configuration (callers with a fallback catch exactly this)."""


def check(rc, name):
    if rc != 0:
        what = _ERR.get(rc) or ('RCCL ncclResult_t %d' % (-16 - rc) if rc <= -16 else 'hipError_t %d' % rc)
        raise (Unsupported if rc == -3 else RuntimeError)('%s failed: %s' % (name, what))


CALL_HOOK = None        # measurement aid (bench.py): ``hook(fn, args, name) -> rc`` runs every C-ABI call, eager or replayed from a launch plan


def call(name, *args):
    """Invoke a C-ABI entry point and raise on a non-zero return."""
    fn = getattr(load(), name)
    check(fn(*args) if CALL_HOOK is None else CALL_HOOK(fn, args, name), name)
