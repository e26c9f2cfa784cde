"""ctypes binding of libmsmp_pde.so (include/msmp_pde.h).  There is no fallback: if the library is
missing or a call fails, this raises."""
import ctypes
import os
from ctypes import c_int, c_int64, c_size_t, c_void_p, c_float, c_double, c_char_p

PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('MSMP_LIB_PATH') or os.path.join(PKG, 'libmsmp_pde.so')     # MSMP_LIB_PATH: a diagnostic build of the same library

class MsmpTiles(ctypes.Structure):
    """msmp_tiles_t (include/msmp_pde.h): node tiles of the LDS-staged message kernel."""
    _fields_ = [('tile_nodes', ctypes.c_int32), ('group_nodes', ctypes.c_int32), ('n_tiles', ctypes.c_int32), ('tile_node', c_void_p), ('tile_count', c_void_p),
                ('tile_halo', c_void_p), ('edge_slot', c_void_p), ('listed', ctypes.c_int32), ('period_tiles', ctypes.c_int32), ('period_nodes', ctypes.c_int32)]


class MsmpDecoder(ctypes.Structure):
    """msmp_decoder_t (include/msmp_pde.h): the 1-D decoder as the epilogue of the last layer's node tail."""
    _fields_ = [('w1', c_void_p), ('b1', c_void_p), ('w2', c_void_p), ('b2', c_void_p), ('u', c_void_p), ('dt', ctypes.c_float),
                ('time_window', ctypes.c_int32), ('out', c_void_p)]


MSMP_TILE_NCAP = 32
MSMP_TILE_EDGES = 128
MSMP_TILE_GROUP_EDGES = 32      # edge lanes of one wave group of a tile
MSMP_LAYER_RESIDUAL_SWISH = 0
MSMP_LAYER_LIN = 1
MSMP_ERR_UNSUPPORTED = -2
MSMP_MAX_VARS = 8
HIDDEN = 128
MSMP_ABI_VERSION = 400          # include/msmp_pde.h: the library must report exactly this (msmp_tiles_t grew in round 4)
MSMP_STATUS_INPUT_RANGE, MSMP_STATUS_NODE_SATURATED, MSMP_STATUS_NONFINITE = 1, 2, 4

# name -> (restype, argtypes); must list every symbol include/msmp_pde.h declares
SIGNATURES = {
    'msmp_version': (c_int, []),
    'msmp_last_error': (c_char_p, []),
    'msmp_last_status': (c_int, [ctypes.POINTER(c_int), c_int]),
    'msmp_tune': (c_int, [c_char_p, c_int]),
    'msmp_tune_query': (c_int, [c_char_p]),
    'msmp_packed_layer_floats': (c_int64, [c_int, c_int]),
    'msmp_pack_layer_f32': (c_int, [c_void_p] * 8 + [c_int, c_int, c_void_p, c_void_p]),
    'msmp_build_csr_workspace_bytes': (c_size_t, [c_int64, c_int64]),
    'msmp_build_csr': (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    'msmp_radius_graph_count_f64': (c_int, [c_void_p, c_int, c_void_p, c_int64, c_int64, c_double, c_int, c_void_p, c_void_p]),
    'msmp_radius_graph_fill_f64': (c_int, [c_void_p, c_int, c_void_p, c_int64, c_int64, c_double, c_int, c_void_p, c_int64,
                                           c_void_p, c_void_p]),
    'msmp_knn_graph_f64': (c_int, [c_void_p, c_int, c_void_p, c_int64, c_int64, c_int, c_int64, c_void_p, c_void_p, c_void_p]),
    'msmp_edge_mlp_f32': (c_int, [c_void_p] * 6 + [c_int64, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    'msmp_scatter_mean_f32': (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    'msmp_node_update_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    'msmp_instance_norm_f32': (c_int, [c_void_p, c_void_p, c_int64, c_int, c_float, c_void_p, c_void_p]),
    'msmp_gate_blend_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_float, c_void_p, c_void_p]),
    'msmp_edge_aggregate_f32': (c_int, [c_void_p] * 7 + [c_int64, c_int64, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    'msmp_node_project_f32': (c_int, [c_void_p] * 4 + [c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    'msmp_edge_aggregate_projected_f32': (c_int, [c_void_p] * 5 + [c_int64, c_int64, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    'msmp_build_tiles': (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'msmp_node_feature_stride': (c_int, [c_int, c_int]),
    'msmp_prepare_nodes': (c_int, [c_void_p, c_int, c_void_p, c_int, c_int64, c_int, c_double, c_double, c_int, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'msmp_pack_node_features_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p]),
    'msmp_edge_aggregate_tiled_f32': (c_int, [c_void_p] * 8 + [ctypes.POINTER(MsmpTiles), c_int64, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    'msmp_mp_layer_workspace_bytes': (c_size_t, [c_int64, c_int64, c_int, c_int]),
    'msmp_mp_layer_f32': (c_int, [c_void_p] * 8 + [ctypes.POINTER(MsmpTiles), c_void_p] + [c_int64, c_int64, c_int64, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int,
                                                   c_float, c_void_p, c_void_p, c_size_t, c_void_p]),
    'msmp_mp_layer_decode_f32': (c_int, [c_void_p] * 8 + [ctypes.POINTER(MsmpTiles), c_void_p] + [c_int64, c_int64, c_int64, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int,
                                                          c_float, c_void_p, ctypes.POINTER(MsmpDecoder), c_void_p, c_size_t, c_void_p]),
    'msmp_packed_lem_floats': (c_int64, []),
    'msmp_pack_lem_f32': (c_int, [c_void_p] * 8 + [c_int, c_void_p, c_void_p]),
    'msmp_lem_input_stride': (c_int, [c_int]),
    'msmp_lem_encoder_f32': (c_int, [c_void_p, c_int64, c_int, c_int, c_float, c_void_p, c_int, c_void_p, c_void_p]),
    'msmp_decoder_f32': (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p]),
    'msmp_decoder2d_f32': (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p]),
    'msmp_node_tail_f32': (c_int, [c_void_p] * 5 + [c_int64, c_int64, c_int, c_int, c_void_p, c_void_p, c_int, c_float, c_void_p, c_void_p]),
    'msmp_lem_encoder_nodes_f32': (c_int, [c_void_p] * 5 + [c_int64, c_int, c_int, c_int, c_float, c_void_p, c_int, c_void_p, c_void_p]),
    'msmp_lem_saved_floats': (c_int64, [c_int64, c_int]),
    'msmp_lem_train_fwd_f32': (c_int, [c_void_p, c_int64, c_int, c_int, c_float] + [c_void_p] * 7),
    'msmp_packed_lem_bwd_floats': (c_int64, []),
    'msmp_pack_lem_bwd_f32': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    'msmp_lem_train_bwd_f32': (c_int, [c_void_p] * 4 + [c_int64, c_int, c_float, c_void_p, c_void_p, c_void_p]),
    'msmp_edge_concat_f32': (c_int, [c_void_p] * 6 + [c_int64, c_int, c_int, c_int, c_void_p, c_void_p]),
    'msmp_mean_bwd_dswish_f32': (c_int, [c_void_p] * 4 + [c_int64, c_void_p, c_void_p]),
    'msmp_instance_norm_bwd_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_float, c_void_p, c_void_p]),
    'msmp_gate_blend_bwd_f32': (c_int, [c_void_p] * 5 + [c_int64, c_float, c_void_p, c_void_p, c_void_p, c_void_p]),
    'msmp_grad_weights_workspace_floats': (c_int64, [c_int, c_void_p, c_void_p]),
    'msmp_grad_weights_f32': (c_int, [c_int] + [c_void_p] * 9 + [c_int64, c_void_p]),
    'msmp_grad_weights_cat_f32': (c_int, [c_int] + [c_void_p] * 12 + [c_int64, c_void_p]),
    'msmp_mp_layer_bwd_workspace_bytes': (c_size_t, [c_int64, c_int64, c_int, c_int, c_int]),
    'msmp_mp_layer_bwd_f32': (c_int, [c_void_p] * 11 + [c_int64, c_int64, c_int64, c_int, c_int, c_void_p, c_void_p, c_int, c_float,
                                      c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    'msmp_adamw_f32': (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float, c_float, c_float, c_float, c_int64, c_void_p]),
    'msmp_adamw_capturable_f32': (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float, c_float, c_float, c_void_p, c_void_p]),
    'msmp_reduce_workspace_bytes': (c_size_t, [c_int]),
    'msmp_colsum_f32': (c_int, [c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    'msmp_sqerr_sum_f32': (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_size_t, c_void_p]),
    'msmp_packed_mlp2_floats': (c_int64, [c_int]),
    'msmp_mlp2_input_stride': (c_int, [c_int]),
    'msmp_pack_mlp2_f32': (c_int, [c_void_p] * 4 + [c_int, c_void_p, c_void_p]),
    'msmp_mlp2_swish_f32': (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p]),
    'msmp_linear_swish_workspace_bytes': (c_size_t, [c_int, c_int]),
    'msmp_linear_swish_f32': (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    'msmp_linear_workspace_bytes': (c_size_t, [c_int, c_int]),
    'msmp_linear_f32': (c_int, [c_void_p, c_int, c_int64, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    'msmp_wide_gather_swish_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p]),
    'msmp_wide_scatter_mean_f32': (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p]),
    'msmp_wide_swish_f32': (c_int, [c_void_p, c_int64, c_void_p, c_void_p]),
    'msmp_wide_lem_z_f32': (c_int, [c_void_p, c_int64, c_int, c_float, c_void_p, c_void_p, c_void_p]),
    'msmp_wide_lem_y_f32': (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    'msmp_wide_norm_blend_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_float, c_void_p, c_void_p]),
    'msmp_timing_enable': (c_int, [c_int]),
    'msmp_timing_reset': (c_int, []),
    'msmp_timing_read': (c_int, [c_int, ctypes.POINTER(c_int64), ctypes.POINTER(c_double)]),
}

K_EDGE_MLP, K_SCATTER_MEAN, K_NODE_UPDATE, K_NORM, K_LEM, K_NODE_PROJ, K_DECODER = 0, 1, 2, 3, 4, 5, 6
MSMP_LAYER_DENSE_MESSAGE = 16

_lib = None


class MsmpError(RuntimeError):
    pass


def lib():
    """The loaded library.  Raises (never falls back) if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MsmpError(f'{LIB_PATH} not found: build it with `python msmp-pde_amd/build.py` '
                            '(or __graft_entry__.build()); there is no CPU fallback')
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)        # AttributeError if the .so lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        if handle.msmp_version() != MSMP_ABI_VERSION:
            raise MsmpError(f'{LIB_PATH} reports ABI version {handle.msmp_version()}, this binding was written for {MSMP_ABI_VERSION}: '
                            'rebuild it with `python msmp-pde_amd/build.py`')
        _lib = handle
    return _lib


def check(rc, what=''):
    if rc != 0:
        msg = lib().msmp_last_error()
        raise MsmpError(f'{what} failed with code {rc}: {msg.decode() if msg else ""}')


def ptr(t):
    """Device (or host) address of a torch tensor, None -> NULL."""
    return None if t is None else t.data_ptr()


# ---- freshness of the packed-weight caches ------------------------------------------------------------------------------
# The kernel-layout weight blobs are cached per module and re-packed when a parameter changed.  `Tensor._version` alone does
# not see every change: fused optimizers (torch.optim.AdamW(fused=True): torch._fused_adamw_) update the parameters without
# bumping it, and so do edits through `.data`.  Every cache key therefore also carries PARAM_EPOCH, which a process-wide
# optimizer hook advances after ANY optimizer.step(); `invalidate_packed_weights()` does the same by hand for exotic in-place
# edits (p.data.copy_(...) and the like).
PARAM_EPOCH = [0]


def invalidate_packed_weights(*_args, **_kwargs):
    PARAM_EPOCH[0] += 1


def _install_optimizer_hook():
    try:
        from torch.optim.optimizer import register_optimizer_step_post_hook
        register_optimizer_step_post_hook(invalidate_packed_weights)
    except Exception as exc:          # very old torch: fall back to version keys only, loudly
        import warnings
        warnings.warn(f'msmp_pde_amd: no global optimizer hook ({exc}); call invalidate_packed_weights() after fused optimizer steps')


_install_optimizer_hook()

_raw_stream = None


def current_stream():
    """Raw hipStream_t of torch's current stream on the current device (every launch is stream-ordered with torch's own work).
    Goes through the C accessor: building a torch.cuda.Stream object per kernel launch costs ~10 us of host time, which is
    what bounds small batches (the launches of a 32-graph rollout step, the backward of a 16-graph training batch)."""
    global _raw_stream
    import torch
    if _raw_stream is None:
        _raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', False)
    if _raw_stream:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def timing_read(kernel):
    """(launches, total_ms) of one kernel family since the last msmp_timing_reset."""
    n, ms = c_int64(0), c_double(0.0)
    check(lib().msmp_timing_read(kernel, ctypes.byref(n), ctypes.byref(ms)), 'msmp_timing_read')
    return n.value, ms.value


# ---- range status of the fp16-split path (msmp_last_status) ----------------------------------------------------------------
class MsmpRangeWarning(RuntimeWarning):
    """A value left the range the default (fp16-split) kernels represent: results are saturated or not finite."""


_STATUS_TEXT = {
    MSMP_STATUS_INPUT_RANGE: 'an input feature (u, pos / L or a variables column) has |x| > 255 or is not finite',
    MSMP_STATUS_NODE_SATURATED: 'a hidden-state / aggregate row has |x| > 255 or is NaN (saturated at 255.87 in the GEMM operands)',
    MSMP_STATUS_NONFINITE: 'the InstanceNorm statistics of a graph are not finite (an activation above 1023 overflowed fp16 upstream)',
}
_status_seen = [0]


def last_status(reset=False):
    """Sticky range flags (MSMP_STATUS_*) of all kernel work that has COMPLETED so far; no device synchronisation."""
    flags = c_int(0)
    check(lib().msmp_last_status(ctypes.byref(flags), int(bool(reset))), 'msmp_last_status')
    if reset:
        _status_seen[0] = 0
    return flags.value


def status_check():
    """One host read of the sticky flags.  New flags -> MsmpRangeWarning (MSMP_STRICT_RANGE=1 in the environment: MsmpError).
    The flags stay set until last_status(reset=True).  Solver.forward no longer stops at the warning: see solvers._SolverBase.forward
    (range_policy) -- a flagged forward is evaluated again on the exact-fp32 kernels."""
    flags = last_status()
    new = flags & ~_status_seen[0]
    if new:
        _status_seen[0] |= new
        _range_report(new, "Outputs since then are saturated / not finite.  Rescale the data or select the exact-fp32 kernels with "
                           "lib().msmp_tune(b'split', 0); clear with last_status(reset=True).")
    return flags


def _range_report(flags, what_now):
    msg = ('msmp_pde_amd: the fp16-split matrix path left its range: ' + '; '.join(t for b, t in _STATUS_TEXT.items() if flags & b)
           + '.  ' + what_now)
    if os.environ.get('MSMP_STRICT_RANGE') == '1':
        raise MsmpError(msg)
    import warnings
    warnings.warn(msg, MsmpRangeWarning, stacklevel=4)


class exact_fp32:
    """Context manager: the kernels launched inside use the exact-fp32 MFMA path (msmp_tune("split", 0): no input-range limit,
    about 2x slower), whatever the process-wide setting is; restored on exit."""

    def __enter__(self):
        self._old = lib().msmp_tune_query(b'split')
        lib().msmp_tune(b'split', 0)
        return self

    def __exit__(self, *exc):
        lib().msmp_tune(b'split', self._old)
        return False
