"""ctypes loader for libdvm_hip.so (C ABI: include/dvm.h).

There is deliberately no CPU fallback: if the shared library is missing, or a
tensor is not on a HIP device, every op raises.  Build with
`python -c "import __graft_entry__ as g; g.build()"` or `make -C dv-matcher_amd/csrc`.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(_HERE), "csrc")
SO_PATH = os.path.join(CSRC, "libdvm_hip.so")

_lib = None

c_int, c_float, c_size_t, c_void_p = ctypes.c_int, ctypes.c_float, ctypes.c_size_t, ctypes.c_void_p

# name -> (restype, [argtypes])  — mirrors include/dvm.h one to one
_P = c_void_p
SIGNATURES = {
    "dvm_abi_version": (c_int, []),
    "dvm_last_error": (ctypes.c_char_p, []),
    "dvm_device_count": (c_int, []),
    "dvm_profile_enable": (c_int, [c_int]),
    "dvm_profile_select": (c_int, [ctypes.c_uint]),
    "dvm_profile_read": (c_int, [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(c_int)]),
    "dvm_profile_read_kernel": (c_int, [c_int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(c_int)]),
    "dvm_profile_kernel_name": (ctypes.c_char_p, [c_int]),
    "dvm_profile_disable": (c_int, []),
    "dvm_rownorm2_f32": (c_int, [_P, c_int, c_int, _P, _P]),
    "dvm_linear_f32": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, c_float, _P, _P]),
    "dvm_linear_prefix_f32": (c_int, [_P, c_int, _P, _P, c_int, c_int, c_int, c_int, _P, _P, _P, _P, c_float, _P, _P]),
    "dvm_linear_scaled_residual_f32": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, _P, c_float, _P, _P, _P]),
    "dvm_linear_wgrad_f32": (c_int, [_P, _P, ctypes.c_long, c_int, c_int, _P, _P]),
    "dvm_bn_pm_workspace_bytes": (c_size_t, [ctypes.c_long, c_int]),
    "dvm_bn_act_train_fwd_pm_f32": (c_int, [_P, _P, _P, _P, ctypes.c_long, c_int, c_float, c_float, c_float, _P, _P, _P, _P, _P, _P, c_size_t, _P]),
    "dvm_bn_act_train_bwd_pm_f32": (c_int, [_P, _P, _P, _P, _P, _P, _P, ctypes.c_long, c_int, c_float, _P, _P, _P, c_int, _P, c_size_t, _P]),
    "dvm_softcorr_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "dvm_softcorr_fwd_f32": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_float, c_int, _P, _P, _P, _P, c_int, _P,
                                     c_size_t, _P]),
    "dvm_softcorr_bwd_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "dvm_softcorr_bwd_f32": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_float, c_int, _P, _P, _P, _P, _P, _P, _P, c_int,
                                     _P, c_size_t, _P]),
    "dvm_n2p_core_fwd_f32": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P, _P]),
    "dvm_n2p_core_bwd_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "dvm_n2p_core_bwd_f32": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P, c_size_t, _P]),
    "dvm_softcorr_apply_bwd_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "dvm_softcorr_apply_bwd_f32": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, c_size_t, _P]),
    "dvm_dist_loss_bwd_weights_f32": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P]),
    "dvm_rot6d_bwd_f32": (c_int, [_P, _P, c_int, _P, _P]),
    "dvm_dg_warp_arap_bwd_f32": (c_int, [_P, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "dvm_chamfer_bwd_f32": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, _P, _P, _P]),
    "dvm_sa_attention_train_fwd_workspace_bytes": (c_size_t, [c_int, c_int]),
    "dvm_sa_attention_train_fwd_f32": (c_int, [_P, _P, c_int, c_int, _P, _P, _P, _P, c_size_t, _P]),
    "dvm_sa_attention_bwd_workspace_bytes": (c_size_t, [c_int, c_int]),
    "dvm_sa_attention_bwd_f32": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, _P, _P, _P, c_size_t, _P]),
    "dvm_pair_set_overlap": (c_int, [c_int]),
    "dvm_pair_init": (c_int, [_P]),
    "dvm_uni3fc_fwd_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "dvm_uni3fc_fwd_f32": (c_int, [_P, _P, c_int, c_int, _P, c_int, c_int, _P, _P, _P, c_size_t, _P]),
    "dvm_uni3fc_train_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "dvm_uni3fc_train_fwd_f32": (c_int, [_P, _P, c_int, c_int, _P, c_int, c_int, c_float, c_float, c_int, c_int, _P, _P, _P, _P, _P, c_size_t, _P]),
    "dvm_uni3fc_train_running_stats_f32": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, c_float, _P, c_size_t, _P]),
    "dvm_bn_act_train_fwd_pm_var_f32": (c_int, [_P, _P, _P, _P, ctypes.c_long, c_int, c_int, c_float, c_float, c_float, _P, _P, _P, _P, _P, _P, _P, c_size_t, _P]),
    "dvm_bn_pm_groups_workspace_bytes": (c_size_t, [ctypes.c_long, c_int, c_int]),
    "dvm_bn_act_train_bwd_pm_groups_f32": (c_int, [_P, _P, _P, _P, _P, _P, _P, ctypes.c_long, c_int, c_int, c_float, _P, _P, _P, c_int, _P, c_size_t, _P]),
    "dvm_uni3fc_train_bwd_f32": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, _P, _P, c_int, c_int, c_int, _P, c_size_t, _P]),
    # the same calls with the caller's collective for cross-rank statistics (dvm_collective *: passed as a pointer)
    "dvm_uni3fc_train_fwd_sync_f32": (c_int, [_P, _P, c_int, c_int, _P, c_int, c_int, c_float, c_float, c_int, c_int, _P, _P, _P, _P, _P, c_size_t, _P, _P]),
    "dvm_uni3fc_train_bwd_sync_f32": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, _P, _P, c_int, c_int, c_int, _P, c_size_t, _P, _P]),
    "dvm_bn_pm_sync_bytes": (c_size_t, [c_int, c_int]),
    "dvm_bn_act_train_fwd_pm_sync_f32": (c_int, [_P, _P, _P, _P, ctypes.c_long, c_int, c_int, c_float, c_float, c_float, _P, _P, _P, _P, _P, _P, _P, c_size_t, _P, _P, _P]),
    "dvm_bn_act_train_bwd_pm_sync_f32": (c_int, [_P, _P, _P, _P, _P, _P, _P, ctypes.c_long, c_int, c_int, c_float, _P, _P, _P, c_int, _P, c_size_t, _P, _P, _P]),
    "dvm_pos_encoding_sync_f32": (c_int, [_P, c_int, c_int, _P, _P, c_size_t, _P, _P, _P]),
    "dvm_criterion_train_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "dvm_criterion_train_fwd_f32": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_float, _P, c_int, c_int, _P, _P, _P, _P, c_int,
                                            c_int, _P, _P, c_size_t, _P]),
    "dvm_criterion_train_bwd_f32": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_float, _P, _P, c_int, c_int, _P, _P, c_int,
                                            c_int, _P, _P, c_size_t, _P]),
    "dvm_criterion_dir_train_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "dvm_criterion_dir_train_fwd_f32": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_float, _P, c_int, c_int, _P,
                                                _P, c_size_t, _P]),
    "dvm_criterion_dir_train_bwd_f32": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_float, _P, _P, c_int,
                                                c_int, _P, _P, _P, c_size_t, _P]),
    "dvm_linear_wgrad_workspace_bytes": (c_size_t, [ctypes.c_long, c_int, c_int]),
    "dvm_linear_wgrad_ws_f32": (c_int, [_P, _P, ctypes.c_long, c_int, c_int, _P, _P, c_size_t, _P]),
    "dvm_set_deterministic": (c_int, [c_int]),
    "dvm_get_deterministic": (c_int, []),
    "dvm_k1_last_routes": (c_int, [_P]),
    "dvm_pair_destroy": (c_int, []),
    "dvm_argmin_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "dvm_argmin_exact_f32": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P, _P, _P, c_size_t, _P]),
    "dvm_argmin_pair_f32": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, c_size_t, _P]),
    "dvm_knn_cdist_workspace_bytes": (c_size_t, [c_int] * 4),
    "dvm_knn_cdist_f32": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P, c_size_t, _P]),
    "dvm_knn_neg_workspace_bytes": (c_size_t, [c_int] * 5),
    "dvm_knn_neg_f32": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P, c_size_t, _P]),
    "dvm_softcorr_dense_workspace_bytes": (c_size_t, [c_int] * 4),
    "dvm_softcorr_dense_f32": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_float, _P, _P, c_size_t, _P]),
    "dvm_deformer_mlp_workspace_bytes": (c_size_t, [c_int]),
    "dvm_deformer_mlp_fwd_f32": (c_int, [_P, c_int] + [_P] * 8 + [_P, _P, c_size_t, _P]),
    "dvm_pos_encoding_workspace_bytes": (c_size_t, []),
    "dvm_pos_encoding_f32": (c_int, [_P, c_int, c_int, _P, _P, c_size_t, _P]),
    "dvm_pos_encoding_minmax_f32": (c_int, [_P, _P, c_int, c_int, _P, _P]),
    "dvm_bn_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "dvm_bn_act_train_fwd_f32": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_float, c_float, c_float, _P, _P, _P, _P, _P, _P,
                                         c_size_t, _P]),
    "dvm_bn_act_train_bwd_f32": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_float, _P, _P, _P, _P, c_size_t, _P]),
    "dvm_graph_geodesics_f64": (c_int, [_P, _P, c_int, c_int, _P, _P]),
    "dvm_proj2img_workspace_bytes": (c_size_t, [c_int]),
    "dvm_proj2img_f32": (c_int, [_P, c_int, c_int, _P, _P, _P, _P, _P, c_size_t, _P]),
    "dvm_i2p_f32": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P, c_int, _P]),
    "dvm_adaptive_conv_f32": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P]),
    "dvm_bicubic_resize_pad_f32": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P]),
    "dvm_jbu_kernel_f32": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P]),
    "dvm_sa_attention_workspace_bytes": (c_size_t, [c_int, c_int]),
    "dvm_sa_attention_fwd_f32": (c_int, [_P, _P, c_int, c_int, _P, _P, c_size_t, _P]),
    "dvm_n2p_attention_fwd_f32": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P]),
    "dvm_dist_loss_workspace_bytes": (c_size_t, [c_int] * 5),
    "dvm_dist_loss_fwd_f32": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, c_size_t, _P]),
    "dvm_softcorr_apply_f32": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P]),
    "dvm_fps_f32": (c_int, [_P, c_int, c_int, c_int, _P, _P, _P]),
    "dvm_dg_build_workspace_bytes": (c_size_t, [c_int, c_int]),
    "dvm_dg_build_f32": (c_int, [_P, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, _P, c_size_t, _P]),
    "dvm_rot6d_f32": (c_int, [_P, c_int, _P, _P]),
    "dvm_dg_warp_arap_fwd_f32": (c_int, [_P, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "dvm_dg_warp_arap_graph_f32": (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "dvm_chamfer_workspace_bytes": (c_size_t, [c_int] * 3),
    "dvm_chamfer_fwd_f32": (c_int, [_P, _P, c_int, c_int, c_int, _P, _P, _P, _P, _P, c_size_t, _P]),
    "dvm_deformer_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "dvm_deformer_fwd_f32": (c_int, [_P] * 9 + [c_int] * 6 + [_P] * 10 + [_P, c_int, _P, c_size_t, _P]),
    "dvm_map_term_workspace_bytes": (c_size_t, [c_int, c_int]),
    "dvm_map_term_f32": (c_int, [_P] * 6 + [c_int] * 5 + [_P, _P, c_size_t, _P]),
    "dvm_pair_direction_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "dvm_pair_direction_fwd_f32": (c_int, [_P] * 4 + [c_int] * 3 + [c_float, _P] + [_P] * 10 + [c_int] + [_P] * 4 +
                                   [_P, c_size_t, _P]),
    "dvm_pair_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "dvm_pair_fwd_f32": (c_int, [_P] * 4 + [c_int] * 3 + [c_float, _P, _P] + [_P] * 10 + [c_int] + [_P] * 8 +
                         [_P, c_size_t, _P]),
    "dvm_pair_fwd_cached_f32": (c_int, [_P] * 4 + [c_int] * 3 + [c_float, _P, _P] + [_P] * 10 + [c_int] + [_P] * 8 +
                                [_P, c_size_t, c_int, _P]),
    "dvm_pair_geometry_f32": (c_int, [_P, _P] + [c_int] * 3 + [_P, _P, c_int, _P, c_size_t, _P]),
}


class DvmError(RuntimeError):
    pass


def load():
    """Load the shared library (once). Raises DvmError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise DvmError("libdvm_hip.so not found at %s — build it first (make -C %s); there is no CPU fallback"
                       % (SO_PATH, CSRC))
    lib = ctypes.CDLL(SO_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here == ABI mismatch, let it propagate
        fn.restype = res
        fn.argtypes = args
    if lib.dvm_abi_version() != 1:
        raise DvmError("libdvm_hip.so ABI version %d != 1" % lib.dvm_abi_version())
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().dvm_last_error()
        raise DvmError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else "?"))
