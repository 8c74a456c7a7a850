"""ctypes binding of libmuzero_hip.so (the C ABI declared in include/muzero_hip.h).

The library is built in-tree by `make -C muzero_amd/csrc` (or __graft_entry__.build()).
There is no CPU fallback: every product path that needs a kernel calls `lib()` and raises
when the shared object is missing or a call fails.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MUZERO_HIP_LIB", os.path.join(_HERE, "libmuzero_hip.so"))
MAX_LAYERS = 8

ACT = {"none": 0, "linear": 0, "relu": 1, "tanh": 2, "sigmoid": 3, "softmax": 4}
SELECT_ROOT_F32 = 1


class HipError(RuntimeError):
    pass


class Pool(C.Structure):
    _fields_ = [("G", C.c_int32), ("A", C.c_int32), ("cap", C.c_int32), ("D", C.c_int32),
                ("node", C.c_void_p), ("n_nodes", C.c_void_p), ("latent", C.c_void_p)]


class Dense(C.Structure):
    _fields_ = [("weight", C.c_void_p), ("bias", C.c_void_p), ("in_features", C.c_int32),
                ("out_features", C.c_int32), ("activation", C.c_int32), ("ldw", C.c_int32)]


class Mlp(C.Structure):
    _fields_ = [("n_layers", C.c_int32), ("layer", Dense * MAX_LAYERS)]


class PackedLayer(C.Structure):
    _fields_ = [("K", C.c_int32), ("N", C.c_int32), ("Kp", C.c_int32), ("Np", C.c_int32),
                ("act", C.c_int32), ("fmt", C.c_int32), ("w_off", C.c_int64), ("b_off", C.c_int64)]


class PackedMlp(C.Structure):
    _fields_ = [("n_layers", C.c_int32), ("in_features", C.c_int32), ("out_features", C.c_int32),
                ("max_width", C.c_int32), ("layer", PackedLayer * MAX_LAYERS)]


class DynamicsNet(C.Structure):
    _fields_ = [("embedding", C.c_void_p), ("merge_weight", C.c_void_p), ("merge_bias", C.c_void_p),
                ("reward", Mlp), ("D", C.c_int32), ("A", C.c_int32)]


class PredictionNet(C.Structure):
    _fields_ = [("value", Mlp), ("policy", Mlp), ("D", C.c_int32), ("A", C.c_int32)]


class C4State(C.Structure):
    _fields_ = [("G", C.c_int32), ("_pad", C.c_int32), ("pieces", C.c_void_p), ("to_move", C.c_void_p),
                ("done", C.c_void_p)]


class PackedNets(C.Structure):
    _fields_ = [("D", C.c_int32), ("A", C.c_int32), ("Dp", C.c_int32), ("_pad", C.c_int32),
                ("total_floats", C.c_int64), ("action_table_off", C.c_int64),
                ("merge", PackedMlp), ("reward", PackedMlp), ("value", PackedMlp), ("policy", PackedMlp)]


class SearchParams(C.Structure):
    _fields_ = [("S", C.c_int32), ("depth", C.c_int32), ("tab_len", C.c_int32), ("lanes_per_tree", C.c_int32),
                ("discount", C.c_double), ("noise_frac", C.c_double), ("sqrt_tab", C.c_void_p),
                ("log_tab", C.c_void_p), ("root_latent", C.c_void_p), ("root_to_play", C.c_void_p),
                ("eta", C.c_void_p), ("rand_child", C.c_void_p), ("leaf_trace", C.c_void_p),
                ("child_trace", C.c_void_p), ("rollout_trace", C.c_void_p), ("prior_trace", C.c_void_p),
                ("stamps", C.c_void_p), ("z_pool", C.c_void_p)]


MAX_BLOCKS = 4


class ResBlock(C.Structure):
    _fields_ = [("conv1_w", C.c_void_p), ("bn1_w", C.c_void_p), ("bn1_b", C.c_void_p), ("conv2_w", C.c_void_p),
                ("bn2_w", C.c_void_p), ("bn2_b", C.c_void_p), ("down_w", C.c_void_p), ("dbn_w", C.c_void_p),
                ("dbn_b", C.c_void_p), ("conv1_bias", C.c_void_p), ("in_channels", C.c_int32), ("out_channels", C.c_int32),
                ("kernel_size", C.c_int32), ("activation", C.c_int32), ("eps1", C.c_float), ("eps2", C.c_float),
                ("epsd", C.c_float), ("_pad", C.c_float)]


class Tower(C.Structure):
    _fields_ = [("n_blocks", C.c_int32), ("C", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
                ("block", ResBlock * MAX_BLOCKS), ("final_w", C.c_void_p), ("final_b", C.c_void_p)]


_LIB = None

_P = C.c_void_p
_SIGS = {
    "mz_last_error": (C.c_char_p, []),
    "mz_abi_version": (C.c_int, []),
    "mz_puct_tables": (C.c_int, [_P, _P, C.c_int32]),
    "mz_tree_init": (C.c_int, [C.POINTER(Pool), _P, _P, _P]),
    "mz_tree_select": (C.c_int, [C.POINTER(Pool), _P, _P, C.c_int32, C.c_int32, _P, _P, _P]),
    "mz_tree_expand_commit": (C.c_int, [C.POINTER(Pool), _P, _P, _P, _P, _P, _P, _P]),
    "mz_tree_select_rows": (C.c_int, [C.POINTER(Pool), _P, _P, C.c_int32, C.c_int32, _P, _P, _P, _P, _P]),
    "mz_tree_commit_backup_select": (C.c_int, [C.POINTER(Pool), _P, _P, _P, _P, _P, _P, _P, C.c_int32, C.c_double, _P, _P, C.c_int32,
                                               C.c_int32, C.c_int32, _P, _P, _P, _P, _P]),
    "mz_tree_select_rows_drawn": (C.c_int, [C.POINTER(Pool), _P, _P, C.c_int32, C.c_int32, _P, _P, _P, _P, _P, _P, _P]),
    "mz_tree_expand_commit_rows": (C.c_int, [C.POINTER(Pool), _P, _P, _P, _P, _P, _P, _P, _P]),
    "mz_tree_root_noise": (C.c_int, [C.POINTER(Pool), _P, C.c_double, _P]),
    "mz_tree_backup": (C.c_int, [C.POINTER(Pool), _P, _P, _P, C.c_int32, C.c_double, _P]),
    "mz_tree_result": (C.c_int, [C.POINTER(Pool), _P, _P, _P, _P, _P]),
    "mz_tree_gather_latent": (C.c_int, [C.POINTER(Pool), _P, _P, _P]),
    "mz_mlp_plan": (C.c_int64, [C.POINTER(Mlp), C.c_int64, C.POINTER(PackedMlp)]),
    "mz_mlp_pack": (C.c_int, [C.POINTER(Mlp), C.POINTER(PackedMlp), _P, _P]),
    "mz_mlp_forward": (C.c_int, [C.POINTER(PackedMlp), _P, _P, C.c_int64, _P, _P, C.c_int64, C.c_int32, _P]),
    "mz_mlp_forward_pair": (C.c_int, [C.POINTER(PackedMlp), _P, C.c_int64, _P, _P, C.c_int64, C.c_int32,
                                      C.POINTER(PackedMlp), _P, C.c_int64, _P, _P, C.c_int64, C.c_int32, _P, _P]),
    "mz_mlp_forward_post": (C.c_int, [C.POINTER(PackedMlp), _P, _P, C.c_int64, _P, _P, C.c_int64, C.c_int32, C.c_int32,
                                      _P, _P]),
    "mz_nets_plan": (C.c_int, [C.POINTER(DynamicsNet), C.POINTER(PredictionNet), C.POINTER(PackedNets)]),
    "mz_nets_pack": (C.c_int, [C.POINTER(DynamicsNet), C.POINTER(PredictionNet), C.POINTER(PackedNets), _P, _P]),
    "mz_dynamics_forward": (C.c_int, [C.POINTER(PackedNets), _P, _P, C.c_int64, _P, _P, _P, C.c_int64, _P, _P,
                                      C.c_int32, _P]),
    "mz_prediction_forward": (C.c_int, [C.POINTER(PackedNets), _P, _P, C.c_int64, _P, C.c_int32, _P, _P, _P,
                                        C.c_int32, _P]),
    "mz_c4_reset": (C.c_int, [C.POINTER(C4State), _P, _P]),
    "mz_c4_step": (C.c_int, [C.POINTER(C4State), _P, _P, _P, _P]),
    "mz_c4_observe": (C.c_int, [C.POINTER(C4State), _P, _P, _P]),
    "mz_c4_move_end": (C.c_int, [C.POINTER(C4State), _P, C.c_int32, _P, _P, _P, C.c_int32, _P, _P, _P, _P, _P]),
    "mz_search_run": (C.c_int, [C.POINTER(Pool), C.POINTER(PackedNets), _P, C.POINTER(SearchParams), _P, _P, _P, _P,
                                _P]),
    "mz_history_push_assemble": (C.c_int, [_P, _P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                           _P, _P]),
    "mz_repr_pointwise": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P]),
    "mz_pick_action": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, _P, _P]),
    "mz_episode_end": (C.c_int, [_P, _P, C.c_int32, C.c_int32, _P, _P]),
    "mz_move_draws": (C.c_int, [C.c_uint64, _P, _P, _P, C.c_float, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P, _P]),
    "mz_gamma_powers": (C.c_int, [C.c_double, C.c_int32, _P]),
    "mz_z_targets": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, _P, _P]),
    "mz_wide_plan": (C.c_int64, [C.POINTER(Dense), C.c_int64, C.POINTER(PackedLayer)]),
    "mz_wide_pack": (C.c_int, [C.POINTER(Dense), C.POINTER(PackedLayer), _P, _P]),
    "mz_wide_workspace": (C.c_int64, [C.POINTER(PackedLayer), C.c_int32]),
    "mz_wide_forward": (C.c_int, [C.POINTER(PackedLayer), _P, _P, C.c_int64, _P, _P, C.c_int64, C.c_int32, _P,
                                  C.c_int64, _P]),
    "mz_wide_static_workspace_floats": (C.c_int64, []),
    "mz_wide_static_init": (C.c_int, []),
    "mz_wide_forward_static": (C.c_int, [C.POINTER(PackedLayer), _P, _P, C.c_int64, _P, _P, C.c_int64, C.c_int32, _P]),
    "mz_nchw_to_nhwc": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, _P, _P]),
    "mz_nhwc_to_nchw": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, _P, _P]),
    "mz_conv_index": (C.c_int, [_P] + [C.c_int32] * 8 + [_P]),
    "mz_conv_weight_kmajor": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, _P, _P]),
    "mz_wide_forward_gather": (C.c_int, [C.POINTER(PackedLayer), _P, _P, C.c_int64, _P, C.c_int32, _P, C.c_int64,
                                         C.c_int32, _P, C.c_int64, _P]),
    "mz_conv_direct_nhwc": (C.c_int, [_P, _P, _P] + [C.c_int32] * 9 + [_P, _P]),
    "mz_bn_stats": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, _P, _P, C.c_float, _P, _P, _P, _P, _P]),
    "mz_bn_apply": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int32, _P, C.c_int32, C.c_int32, C.c_int32, _P]),
    "mz_bn_running_update": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_float, _P, _P, _P]),
    "mz_pool2d_nhwc": (C.c_int, [_P] + [C.c_int32] * 7 + [_P, _P]),
    "mz_adaptive_avg_to_nchw": (C.c_int, [_P] + [C.c_int32] * 6 + [_P, _P]),
    "mz_per_sample": (C.c_int, [_P, C.c_int32, C.c_double, C.c_double, _P, C.c_int32, _P, _P, _P, _P, _P, C.c_int32,
                                C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P, _P, _P]),
    "mz_per_update": (C.c_int, [_P, _P, _P, _P, C.c_int32, C.c_int32, _P]),
    "mz_replay_cut_windows": (C.c_int, [_P, _P, _P, _P, _P] + [C.c_int32] * 5 + [_P, _P] + [C.c_int32] * 4 +
                              [_P, _P, _P, _P, _P, _P, _P]),
    "mz_unroll_loss": (C.c_int, [_P] * 7 + [C.c_int32] * 3 + [C.c_float] * 3 + [_P] * 7),
    "mz_split3": (C.c_int, [_P, C.c_int64, _P, C.c_int64, _P]),
    "mz_nchw_split3": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, _P, C.c_int64, _P]),
    "mz_unsplit3": (C.c_int, [_P, C.c_int64, C.c_int64, _P, _P]),
    "mz_conv3_forward": (C.c_int, [_P, C.c_int64, C.c_int32, _P, C.c_int32, _P, C.c_int64, _P, C.c_int32, _P, _P, C.c_int64,
                                   C.c_int32, C.c_int32, _P, _P]),
    "mz_conv3_forward_f32": (C.c_int, [_P, C.c_int32, _P, C.c_int32, _P, C.c_int64, _P, C.c_int32, _P, _P, C.c_int64,
                                       C.c_int32, C.c_int32, _P, _P]),
    "mz_conv3_same_fits": (C.c_int, [C.c_int32] * 8),
    "mz_conv3_same_block_forward": (C.c_int, [_P] * 6 + [C.c_int32, _P] + [C.c_int32] * 5 + [_P, C.c_int64, _P, C.c_int32, _P, _P, C.c_int64,
                                               C.c_int32, _P, _P]),
    "mz_conv3_same_bn_forward": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P,
                                           C.c_int64, _P, C.c_int32, _P, _P, C.c_int64, C.c_int32, _P, _P]),
    "mz_bn_stats_from_sums": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, _P, _P, C.c_float, _P, _P, _P, _P, _P]),
    "mz_conv3_same_forward": (C.c_int, [_P, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, C.c_int64, _P,
                                        C.c_int32, _P, _P, C.c_int64, C.c_int32, _P, _P]),
    "mz_bn_apply3": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, C.c_int64, C.c_int32, _P, _P, C.c_int64, C.c_int32, C.c_int32,
                               C.c_int32, _P]),
    "mz_merge_conv_forward": (C.c_int, [C.POINTER(PackedLayer), C.POINTER(PackedLayer), _P, _P, C.c_int64, _P, _P, _P, C.c_int32,
                                        C.c_int32, C.c_int32, C.c_int32, _P, _P]),
    "mz_merge_conv_forward_split": (C.c_int, [C.POINTER(PackedLayer), _P, C.POINTER(PackedLayer), _P, _P, C.c_int64, _P, _P, _P,
                                              C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P]),
    "mz_packed_split3_words": (C.c_int64, [C.POINTER(PackedLayer)]),
    "mz_packed_split3": (C.c_int, [C.POINTER(PackedLayer), _P, _P, _P]),
    "mz_tower_thin_min_calls": (C.c_int, [C.c_int32]),
    "mz_search_seq_three_slots": (C.c_int, [C.c_int32]),
    "mz_tower_tail_forward": (C.c_int, [C.POINTER(Tower), _P, C.c_int32, C.c_int32, _P, C.c_int64, _P, _P, _P]),
    "mz_tower_stats_floats": (C.c_int64, [C.POINTER(Tower), C.c_int32]),
    "mz_tower_forward_stats": (C.c_int, [C.POINTER(Tower), _P, C.c_int64, _P, _P, C.c_int64, _P, _P, C.c_int64, _P,
                                         _P, C.c_int64, C.c_int32, C.c_int32, _P, _P]),
    "mz_tower_forward": (C.c_int, [C.POINTER(Tower), _P, C.c_int64, _P, _P, C.c_int64, _P, _P, C.c_int64, _P,
                                   _P, C.c_int64, C.c_int32, C.c_int32, _P]),
    "mz_action_table": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, _P, _P]),
    "mz_policy_post": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P]),
}


def exported_symbols() -> list[str]:
    return sorted(_SIGS)


def register(name: str, restype, argtypes) -> None:
    """Adds a symbol signature (used by sibling modules that bind further entry points)."""
    _SIGS[name] = (restype, argtypes)
    if _LIB is not None:
        fn = getattr(_LIB, name)
        fn.restype, fn.argtypes = restype, argtypes


def lib() -> C.CDLL:
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise HipError(f"{LIB_PATH} is missing: build it with `make -C muzero_amd/csrc` "
                           "(there is no CPU fallback for the search kernels)")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(handle, name)  # AttributeError if the library lacks a declared symbol
            fn.restype, fn.argtypes = res, args
        _LIB = handle
    return _LIB


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise HipError(f"{what} failed ({rc}): {lib().mz_last_error().decode(errors='replace')}")


def ptr(t) -> int | None:
    """Device (or host) address of a torch tensor / numpy array; None passes NULL."""
    if t is None:
        return None
    if hasattr(t, "data_ptr"):
        return t.data_ptr()
    return t.ctypes.data
