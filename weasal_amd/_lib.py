"""ctypes binding of libweasal_hip.so (the C ABI declared in include/weasal_hip.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C weasal_amd/csrc``.
There is NO fallback: if the shared object is missing or a symbol cannot be resolved the
import of any compute path raises, so that a silent CPU/PyTorch substitute can never stand
in for the HIP kernels.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libweasal_hip.so")

_vp = C.c_void_p
_i32 = C.c_int32
_i64 = C.c_int64
_f32 = C.c_float

# name -> (restype, argtypes); mirrors include/weasal_hip.h declaration by declaration
SIGNATURES = {
    "ws_last_error": (C.c_char_p, []),
    "ws_version": (C.c_char_p, []),
    "ws_device_count": (C.c_int, []),
    "ws_kpconv_gather_fwd": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _i32, _vp, _i32, _vp, _i32, _vp, _vp,
                                      _f32, _i32, _i32, _vp, _vp, _vp, _vp]),
    "ws_kpconv_gather_bwd_x": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _i32, _vp, _vp, _vp, _i32, _vp, _i32,
                                        _vp, _vp, _f32, _i32, _i32, _vp, _vp, _vp]),
    "ws_kpconv_gather_bwd_geom": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _i32, _vp, _i32, _vp, _vp, _i32,
                                           _vp, _vp, _vp, _f32, _i32, _i32, _vp, _vp, _vp]),
    "ws_transpose_scratch_bytes": (_i64, [_i64, _i32, _i64]),
    "ws_transpose_build": (C.c_int, [_vp, _i64, _i32, _i64, _vp, _vp, _vp, _vp]),
    "ws_max_pool_fwd": (C.c_int, [_vp, _i64, _i32, _vp, _i64, _i32, _vp, _vp, _vp]),
    "ws_max_pool_bwd": (C.c_int, [_vp, _vp, _i64, _i32, _i32, _vp, _vp, _i64, _vp, _vp]),
    "ws_closest_pool_fwd": (C.c_int, [_vp, _i64, _i32, _vp, _i64, _i32, _vp, _vp]),
    "ws_closest_pool_bwd": (C.c_int, [_vp, _i64, _i32, _i32, _vp, _vp, _i64, _vp, _vp]),
    "ws_gemm_xb": (C.c_int, [_vp, _i64, _i32, _i64, _vp, _i32, _vp, _i64, _vp]),
    "ws_gemm_xb_epilogue": (C.c_int, [_vp, _i64, _i32, _i64, _vp, _i32, _vp, _vp, _i64, _i32, _f32, _vp, _i64, _vp]),
    "ws_act_bwd_colsum_scratch_bytes": (_i64, [_i64, _i32]),
    "ws_act_bwd_colsum": (C.c_int, [_vp, _i64, _i32, _i64, _vp, _i64, C.c_float, _vp, _i64, _vp, _vp, _vp]),
    "ws_contrast_rows_fwd": (C.c_int, [_vp, _i64, _i32, _vp, _i32, _vp, _vp, _vp, C.c_float, C.c_float, _vp, _vp, _vp, _vp, _vp]),
    "ws_contrast_rows_bwd_scratch_bytes": (_i64, [_i64, _i32, _i32]),
    "ws_contrast_rows_bwd": (C.c_int, [_vp, _i64, _i32, _vp, _i32, _vp, _vp, _vp, C.c_float, _vp, _vp, _vp, _vp, _vp, _vp,
                                       _vp, _vp]),
    "ws_launch_count": (_i64, []),
    "ws_radius_neighbors_async_cap": (_i32, [_i32]),
    "ws_radius_neighbors_nearest_async": (C.c_int, [_vp, _vp, _i64, _vp, _i64, _vp, _vp, _i32, C.c_float, _vp, _vp, _vp, _vp]),
    "ws_max_pool_fwd_ordered": (C.c_int, [_vp, _i64, _i32, _vp, _i64, _i32, _vp, _vp, _vp, _vp]),
    "ws_max_pool_bwd_ordered": (C.c_int, [_vp, _vp, _i64, _i32, _i32, _vp, _vp, _i64, _vp, _vp, _vp]),
    "ws_max_pool_fwd_ordered_bf16": (C.c_int, [_vp, _i64, _i32, _vp, _i64, _i32, _vp, _vp, _vp, _vp]),
    "ws_max_pool_bwd_ordered_bf16": (C.c_int, [_vp, _vp, _i64, _i32, _i32, _vp, _vp, _i64, _vp, _vp, _vp]),
    "ws_dropout_apply": (C.c_int, [_vp, _i64, C.c_float, C.c_uint64, _vp, _vp]),
    "ws_gemm_xb_dropout_strided": (C.c_int, [_vp, _i64, _i32, _i64, _vp, _i64, _i64, _i32, _vp, _vp, _i64, _i32, _f32, _f32, C.c_uint64,
                                             _vp, _i64, _vp, _i64, _vp]),
    "ws_gemm_xb_gate_dropout": (C.c_int, [_vp, _i64, _i32, _i64, _vp, _i32, _vp, _i64, _f32, _f32, C.c_uint64, _vp, _i64, _vp, _i64, _vp]),
    "ws_act_bwd_colsum_dropout": (C.c_int, [_vp, _i64, _i32, _i64, _vp, _i64, _f32, _f32, C.c_uint64, _vp, _i64, _vp, _vp, _vp]),
    "ws_pyramid_build": (C.c_int, [_vp, _vp, _vp, _vp]),
    "ws_pyramid_desc_bytes": (_i64, []),
    "ws_contrast_head_scratch_bytes": (_i64, [_i64]),
    "ws_contrast_head_fwd": (C.c_int, [_vp, _i64, _i32, _i64, _vp, C.c_float, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                       _vp, _vp]),
    "ws_contrast_tail_scratch_bytes": (_i64, [_i64]),
    "ws_contrast_tail_fwd": (C.c_int, [_vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ws_contrast_tail_bwd": (C.c_int, [_vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp]),
    "ws_contrast_head_bwd": (C.c_int, [_vp, _vp, _vp, _i32, _vp, _vp, _i64, _i32, _vp, _i64, _vp]),
    "ws_radius_neighbors_set_key_last": (C.c_int, [_vp, _vp]),
    "ws_radius_neighbors_reuse_grid": (C.c_int, [_vp, _i32]),
    "ws_radius_neighbors_grid_info": (C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "ws_radius_neighbors_grid_export": (C.c_int, [_vp, _vp, _vp]),
    "ws_kpconv_gather_bwd_x_grid": (C.c_int, [_vp, _i64, _vp, _i32, _i64, _vp, C.c_float, _vp, _i32, _vp, _i32, _vp, _vp,
                                              C.c_float, _i32, _i32, _vp, _vp, _vp, _vp]),
    "ws_gemm_xb_scratch_bytes": (_i64, [_i64, _i32, _i32]),
    "ws_gemm_xb_epilogue_splitk": (C.c_int, [_vp, _i64, _i32, _i64, _vp, _i32, _vp, _vp, _i64, _i32, C.c_float, _vp, _i64,
                                             _vp, _i64, _vp]),
    "ws_gemm_xty_scratch_bytes": (_i64, [_i64, _i32, _i32]),
    "ws_gemm_xty": (C.c_int, [_vp, _i64, _i32, _i64, _vp, _i32, _i64, _vp, _vp, _vp]),
    "ws_neighbors_ws_create": (C.c_int, [C.POINTER(_vp)]),
    "ws_neighbors_ws_destroy": (None, [_vp]),
    "ws_radius_neighbors_plan": (C.c_int, [_vp, _vp, _i64, _vp, _i64, _vp, _vp, _i32, _f32,
                                          C.POINTER(_i32), _vp]),
    "ws_radius_neighbors_fill": (C.c_int, [_vp, _i32, _vp, _vp, _vp]),
    "ws_radius_neighbors_search": (C.c_int, [_vp, _vp, _i64, _vp, _i64, _vp, _vp, _i32, _f32, _i32, _vp, _vp,
                                            C.POINTER(_i32), _vp]),
    "ws_radius_neighbors_search_async": (C.c_int, [_vp, _vp, _i64, _vp, _i64, _vp, _vp, _i32, _f32, _i32, _vp, _vp,
                                                  _vp, _vp]),
    "ws_radius_neighbors_order": (C.c_int, [_vp, _vp, _vp]),
    "ws_radius_neighbors_counts": (_vp, [_vp]),
    "ws_subsample_ws_create": (C.c_int, [C.POINTER(_vp)]),
    "ws_subsample_ws_destroy": (None, [_vp]),
    "ws_grid_subsample_plan": (C.c_int, [_vp, _vp, _i64, _vp, _i32, _f32, _i32, _i32, _vp,
                                        C.POINTER(_i64), _vp]),
    "ws_grid_subsample_fill": (C.c_int, [_vp, _vp, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ws_rotate_clouds": (C.c_int, [_vp, _i64, _vp, _i32, _vp, _i32, _vp, _vp]),
    "ws_rotate_clouds_host": (C.c_int, [_vp, _i64, _vp, _i32, _vp, _i32, _vp, _vp]),
    # bf16-feature path (same argument lists as the f32 entries; uint16_t* rows)
    "ws_kpconv_gather_fwd_bf16": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _i32, _vp, _i32, _vp, _i32, _vp, _vp,
                                           _f32, _i32, _i32, _vp, _vp, _vp, _vp]),
    "ws_kpconv_gather_bwd_x_bf16": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _i32, _vp, _vp, _vp, _i32, _vp, _i32,
                                             _vp, _vp, _f32, _i32, _i32, _vp, _vp, _vp]),
    "ws_kpconv_gather_bwd_geom_bf16": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _i32, _vp, _i32, _vp, _vp, _i32,
                                                _vp, _vp, _vp, _f32, _i32, _i32, _vp, _vp, _vp]),
    "ws_kpconv_gather_bwd_x_grid_bf16": (C.c_int, [_vp, _i64, _vp, _i32, _i64, _vp, C.c_float, _vp, _i32, _vp, _i32, _vp, _vp,
                                                   C.c_float, _i32, _i32, _vp, _vp, _vp, _vp]),
    "ws_max_pool_fwd_bf16": (C.c_int, [_vp, _i64, _i32, _vp, _i64, _i32, _vp, _vp, _vp]),
    "ws_max_pool_bwd_bf16": (C.c_int, [_vp, _vp, _i64, _i32, _i32, _vp, _vp, _i64, _vp, _vp]),
    "ws_closest_pool_fwd_bf16": (C.c_int, [_vp, _i64, _i32, _vp, _i64, _i32, _vp, _vp]),
    "ws_closest_pool_bwd_bf16": (C.c_int, [_vp, _i64, _i32, _i32, _vp, _vp, _i64, _vp, _vp]),
    "ws_gemm_xbt_bf16": (C.c_int, [_vp, _i64, _i32, _i64, _vp, _i32, _i64, _vp, _vp, _i64, _i32, _f32, _vp, _i64, _i32, _vp]),
    "ws_gemm_xty_bf16": (C.c_int, [_vp, _i64, _i32, _i64, _vp, _i32, _i64, _vp, _vp, _vp]),
    "ws_act_bwd_colsum_bf16_scratch_bytes": (_i64, [_i64, _i32]),
    "ws_act_bwd_colsum_bf16": (C.c_int, [_vp, _i32, _i64, _i32, _i64, _vp, _i64, C.c_float, _vp, _i64, _vp, _vp, _vp]),
    # whole blocks behind one call (descriptor structs: weasal_amd/fused.py mirrors them as ctypes.Structure)
    "ws_kpblock_fwd_scratch_bytes": (_i64, [_vp]),
    "ws_kpblock_bwd_scratch_bytes": (_i64, [_vp]),
    "ws_kpblock_fwd": (C.c_int, [_vp, _vp, _i64, _vp]),
    "ws_kpblock_bwd": (C.c_int, [_vp, _vp, _i64, _vp]),
    "ws_upunary_fwd_scratch_bytes": (_i64, [_vp]),
    "ws_upunary_bwd_scratch_bytes": (_i64, [_vp]),
    "ws_upunary_fwd": (C.c_int, [_vp, _vp, _i64, _vp]),
    "ws_upunary_bwd": (C.c_int, [_vp, _vp, _i64, _vp]),
    "ws_gemm_xb_epilogue_strided": (C.c_int, [_vp, _i64, _i32, _i64, _vp, _i64, _i64, _i32, _vp, _vp, _i64, _i32, _f32, _vp, _i64,
                                              _vp, _i64, _vp]),
    "ws_gemm_xb_gated_strided": (C.c_int, [_vp, _i64, _i32, _i64, _vp, _i64, _i64, _i32, _vp, _vp, _i64, _i32, _f32, _vp, _i64, _f32,
                                           _vp, _i64, _f32, _vp, _i64, _vp, _i64, _vp]),
    "ws_kpconv_gather_bwd_x_gated": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _i32, _vp, _vp, _vp, _i32, _vp, _i32,
                                              _vp, _vp, _f32, _i32, _i32, _vp, _vp, _f32, _vp, _vp]),
    "ws_kpconv_gather_bwd_x_grid_gated": (C.c_int, [_vp, _i64, _vp, _i32, _i64, _vp, C.c_float, _vp, _i32, _vp, _i32, _vp, _vp,
                                                    C.c_float, _i32, _i32, _vp, _vp, _f32, _vp, _i32, _vp, _vp, _vp]),
    "ws_vote_update": (C.c_int, [_vp, _i64, _i32, _vp, _f32, _vp, _vp, _i64, _f32, _vp]),
    "ws_project_confusion": (C.c_int, [_vp, _i32, _vp, _i64, _vp, _vp, _i32, _vp, _vp]),
    "ws_potentials_scratch_bytes": (_i64, [_i64]),
    "ws_potentials_update": (C.c_int, [_vp, _i64, _vp, C.c_double, _vp, _vp, _vp, _vp, _vp]),
    "ws_softmax_ce_scratch_bytes": (_i64, [_i64]),
    "ws_softmax_ce_fwd": (C.c_int, [_vp, _i64, _i32, _i64, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp]),
    "ws_softmax_ce_bwd": (C.c_int, [_vp, _i64, _i32, _i64, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _i64, _vp]),
    "ws_sgd_step": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _f32, _f32, _f32, _f32, _i32, _vp]),
    "ws_kpconv_deform_prepare": (C.c_int, [_vp, _i64, _i32, _vp, _i32, _f32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "ws_kpconv_gather_fwd_ex": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _i32, _vp, _i32, _vp, _i32, _vp, _vp, _f32, _i32, _i32, _vp,
                                         _vp, _vp, _i32, _i32, _vp]),
    "ws_kpconv_deform_prepare_bwd": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _i32, _f32, _i32, _vp, _vp]),
    "ws_kpconv_gather_fwd_def": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _i32, _vp, _i32, _vp, _i32, _f32, _vp, _vp, _vp, _i32, _i32, _vp]),
    "ws_kpconv_gather_bwd_x_def": (C.c_int, [_vp, _i64, _vp, _i64, _i32, _vp, _vp, _vp, _i32, _vp, _i32, _f32, _vp, _vp, _i32, _vp]),
    "ws_kpconv_gather_bwd_x_grid_wide": (C.c_int, [_vp, _i64, _vp, _i32, _i64, _vp, _f32, _vp, _i32, _vp, _i32, _vp, _vp, _f32, _vp,
                                                   _vp, _i32, _vp, _i32, _vp]),
    "ws_kpconv_gather_bwd_geom_def": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _i32, _vp, _i32, _vp, _vp, _i32, _vp, _f32, _vp, _vp,
                                                _i32, _i32, _vp]),
    "ws_kpconv_layer_fwd_fused": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _i32, _vp, _i32, _vp, _i32, _f32, _vp, _vp, _i32, _vp, _i32, _f32,
                                           _vp, _vp]),
    "ws_kpconv_gather_fwd_variant": (C.c_int, [_i32, _i32, _i32, _i32, _i32, _i32, C.c_char_p, _i32]),
    "ws_p2p_regularizer_scratch_bytes": (_i64, [_i64]),
    "ws_p2p_regularizer_fwd": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _f32, _f32, _vp, _vp, _vp]),
    "ws_p2p_regularizer_bwd": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _f32, _f32, _vp, _vp, _vp, _vp]),
    "ws_timer_reset": (C.c_int, []),
    "ws_timer_count": (C.c_int, []),
    "ws_timer_read": (C.c_int, [_i32, _vp, _vp, _vp, _vp]),
    "ws_timer_read_layer": (C.c_int, [_i32, _vp]),
}

_lib = None


class WeasalHipError(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes handle; raises if the HIP library is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise WeasalHipError(
                "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C weasal_amd/csrc` (there is no CPU fallback)" % LIB_PATH)
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)   # AttributeError if the .so is stale
            fn.restype = res
            fn.argtypes = args
        # diagnostics switches of the library (integer globals), settable from the environment for A/B runs
        for env, sym in (("WEASAL_GEMM_SPLIT", "ws_gemm_split"), ("WEASAL_K4G_ROWS", "ws_kpconv_grid_rows"),
                         ("WEASAL_NB_BUCKET", "ws_nb_bucket128"), ("WEASAL_NB_WIDE_CAPS", "ws_nb_wide_caps"), ("WEASAL_NB_MAX_BLOCKS", "ws_nb_max_blocks"), ("WEASAL_POOL_INTERLEAVE", "ws_pool_interleave"), ("WEASAL_CLOSEST_BWD_VEC", "ws_closest_bwd_vec"), ("WEASAL_POOL_SPLIT_ROWS", "ws_pool_split_rows"), ("WEASAL_POOL_UNROLL", "ws_pool_unroll"), ("WEASAL_K4G_INTERLEAVE", "ws_kpconv_grid_interleave"), ("WEASAL_K4_INTERLEAVE", "ws_kpconv_table_interleave"), ("WEASAL_K4GW_INTERLEAVE", "ws_kpconv_gridw_interleave"), ("WEASAL_K6_INTERLEAVE", "ws_kpconv_k6_interleave"), ("WEASAL_K3_SPLIT_ROWS", "ws_kpconv_split_rows"), ("WEASAL_K3_SPLIT_NT", "ws_kpconv_split_nt"), ("WEASAL_GEMM_LOG", "ws_gemm_log"), ("WEASAL_GEMM_STAGED", "ws_gemm_staged"), ("WEASAL_GEMM_THIN_K", "ws_gemm_thin_k"), ("WEASAL_GEMM_SHALLOW", "ws_gemm_shallow"), ("WEASAL_CONTRAST_VARIANT", "ws_contrast_variant")):
            if env in os.environ:
                try:
                    C.c_int.in_dll(handle, sym).value = int(os.environ[env])
                except ValueError:
                    raise WeasalHipError("%s is set but this build of libweasal_hip.so has no switch %s (a lab-only option)" % (env, sym))
        _lib = handle
    return _lib


def check(rc):
    """Map a ws_status to the exception type the reference's modules raise (RuntimeError)."""
    if rc != 0:
        msg = lib().ws_last_error()
        raise WeasalHipError("libweasal_hip status %d: %s" % (rc, msg.decode() if msg else "?"))


def ptr(t):
    """device/host pointer of a tensor (None -> NULL)"""
    return None if t is None else _vp(t.data_ptr())


_raw_stream = None


def current_stream():
    """the calling thread's current HIP stream as a void pointer.  torch.cuda.current_stream() builds a Stream object through
    three Python layers (~12 us per call, ~100 calls per training step: 1 ms of a 3 ms launch-bound step); the raw handle
    comes from one C call."""
    global _raw_stream
    import torch
    if _raw_stream is None:
        get = getattr(torch._C, "_cuda_getCurrentRawStream", None)
        dev = getattr(torch._C, "_cuda_getDevice", None)
        if get is not None and dev is not None and torch.cuda.is_available():
            torch.cuda.current_stream()                          # (initialises the runtime once)
            _raw_stream = lambda: get(dev())
        else:
            _raw_stream = lambda: torch.cuda.current_stream().cuda_stream
    return _vp(_raw_stream())
