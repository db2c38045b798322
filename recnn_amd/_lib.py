"""ctypes binding of librecnn_hip.so (the C ABI declared in include/recnn_hip.h).

The library is the product: there is NO fallback.  If it has not been built
(`python -c "import __graft_entry__ as g; g.build()"` or `make -C recnn_amd/csrc`) every
entry point raises `RecnnHipMissing`.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RECNN_HIP_LIB") or os.path.join(_HERE, "csrc", "librecnn_hip.so")   # override: A/B of builds

F32, BF16, BF16X3 = 0, 1, 2      # BF16X3: split bf16 (hi + lo, three MFMAs per product), include/recnn_hip.h
DTYPES = {"fp32": F32, "bf16": BF16, "bf16x3": BF16X3}
MASK_NONE, MASK_HASH, MASK_EXTERNAL = 0, 1, 2
ALGO_DDPG, ALGO_TD3 = 0, 1
OPT_ADAM, OPT_RANGER = 0, 1
CAT_SOFTMAX, CAT_SAMPLE = 1, 2
LPB_ACCUMULATE, LPB_BF16 = 1, 2
NET_POLICY, NET_TARGET_POLICY, NET_VALUE1, NET_TARGET_VALUE1, NET_VALUE2, NET_TARGET_VALUE2 = range(6)


class RecnnHipMissing(RuntimeError):
    pass


class RecnnHipError(RuntimeError):
    pass


class GemmArgs(C.Structure):
    _fields_ = [
        ("dtype", C.c_int), ("M", C.c_int), ("N", C.c_int),
        ("A", C.c_void_p * 2), ("B", C.c_void_p * 2),
        ("lda", C.c_int64 * 2), ("ldb", C.c_int64 * 2),
        ("K", C.c_int * 2), ("a_f32", C.c_int * 2), ("b_f32", C.c_int * 2),
        ("C", C.c_void_p), ("ldc", C.c_int64), ("c_f32", C.c_int),
        ("bias", C.c_void_p), ("relu", C.c_int), ("mask_mode", C.c_int),
        ("mask", C.c_void_p), ("ld_mask", C.c_int64),
        ("seed", C.c_uint32), ("stream_id", C.c_uint32), ("step_ptr", C.c_void_p),
        ("addend", C.c_void_p), ("ld_add", C.c_int64), ("add_clip", C.c_float), ("add_row_div", C.c_int),
        ("yref", C.c_void_p), ("ldy", C.c_int64), ("dx_scale", C.c_float),
        ("colsum", C.c_void_p),
        ("dw_splits", C.c_int), ("dw_slab_stride", C.c_int64),
        ("dw_valid_cols", C.c_int), ("dw_col_rot", C.c_int),
        ("ws", C.c_void_p), ("ws_bytes", C.c_int64),
    ]


class ShadowOut(C.Structure):
    """include/recnn_hip.h recnn_shadow_out"""
    _fields_ = [("dst", C.c_void_p), ("cols", C.c_int), ("ld", C.c_int64), ("bf16", C.c_int)]


class EngineConfig(C.Structure):
    _fields_ = [
        ("algo", C.c_int), ("dtype", C.c_int), ("state_dim", C.c_int), ("action_dim", C.c_int),
        ("hidden", C.c_int), ("max_rows", C.c_int), ("mask_mode", C.c_int), ("seed", C.c_uint32),
        ("device", C.c_int),
    ]


class Hyper(C.Structure):
    _fields_ = [
        ("gamma", C.c_float), ("min_value", C.c_float), ("max_value", C.c_float),
        ("soft_tau", C.c_float), ("policy_every", C.c_int),
        ("noise_std", C.c_float), ("noise_clip", C.c_float),
        ("lr", C.c_float * 2), ("beta1", C.c_float * 2), ("beta2", C.c_float * 2),
        ("eps", C.c_float * 2), ("weight_decay", C.c_float * 2),
        ("opt_kind", C.c_int * 2), ("la_alpha", C.c_float * 2), ("la_k", C.c_int * 2), ("nsma_threshold", C.c_float * 2),
    ]


class EngineSizes(C.Structure):
    _fields_ = [
        ("master_floats_actor", C.c_int64), ("master_floats_critic", C.c_int64),
        ("workspace_bytes", C.c_int64), ("ld_x", C.c_int64), ("x_rows", C.c_int64),
    ]


class Sampler(C.Structure):
    _fields_ = [
        ("items", C.c_void_p), ("ratings", C.c_void_p), ("user_off", C.c_void_p), ("perm", C.c_void_p),
        ("users_per_batch", C.c_int), ("n_batches", C.c_int), ("frame", C.c_int), ("emb_dim", C.c_int),
        ("table", C.c_void_p), ("row_off", C.c_void_p), ("cursor", C.c_void_p),
        ("plan", C.c_void_p), ("plan_rows", C.c_int),
    ]


TUNING_FIELDS = ("fused_mlp", "chain_target_critic", "bwd_panel", "policy_chain", "split_fwd", "cycle_min_len", "cycle_min_seg",
                 "frozen_fused", "frozen_gemm", "graph_run", "pregather", "defer_policy_fwd", "sampler_f32_rows", "dw_splits", "comm_fused",
                 "l1_big", "gemm_variant", "gemm_v0_threshold", "gemm_dma", "gemm_dma_depth", "gemm_dma_waves", "gemm_waves", "dw_dma", "x3_tail", "x3_fwd", "dw_fuse", "tail_half", "l1_ws", "frozen_half")


class EngineTuning(C.Structure):
    """include/recnn_hip.h recnn_engine_tuning: schedule / tile choices of ONE engine (all compute the same numbers)."""
    _fields_ = [(f, C.c_int) for f in TUNING_FIELDS] + [("reserved", C.c_int * 3)]


_P = C.c_void_p
_I = C.c_int
_L = C.c_int64
_F = C.c_float
_U = C.c_uint32

# name -> (restype, argtypes).  Mirrors include/recnn_hip.h one to one.
SIGNATURES = {
    "recnn_abi_version": (_I, []),
    "recnn_last_error": (C.c_char_p, []),
    "recnn_abi_sizeof": (_L, [_I]),
    "recnn_engine_tuning_init": (None, [C.POINTER(EngineTuning)]),
    "recnn_engine_set_tuning": (_I, [_P, C.POINTER(EngineTuning)]),
    "recnn_engine_get_tuning": (_I, [_P, C.POINTER(EngineTuning)]),
    "recnn_engine_sampler_eager": (_I, [_P, _I]),
    "recnn_engine_unit_backward": (_I, [_P]),
    "recnn_engine_dp_sets": (_I, [_P]),
    "recnn_engine_read_counters": (_I, [_P, _P, _P]),
    "recnn_frame_plan_rows": (_I, [_P, _P, _I, _I, _I, _I, _P, _P]),
    "recnn_frame_plan_dense": (_I, [_P, _P, _I, _I, _I, _I, _P, _L, _P, _P]),
    "recnn_frame_plan": (_I, [_P, _P, _I, _I, _P, _P, _I, _P]),
    "recnn_frame_gather": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _L, _P, _L, _P, _L, _P, _P, _P, _I, _P]),
    "recnn_pack_batch": (_I, [_P, _L, _P, _L, _P, _L, _I, _I, _I, _P, _P, _L, _P]),
    "recnn_gemm_fwd": (_I, [C.POINTER(GemmArgs), _P]),
    "recnn_gemm_dx": (_I, [C.POINTER(GemmArgs), _P]),
    "recnn_gemm_dw": (_I, [C.POINTER(GemmArgs), _P]),
    "recnn_x3_pack": (_I, [_P, _L, _I, _I, _P, _L, _P]),
    "recnn_x3_unpack": (_I, [_P, _L, _I, _I, _P, _L, _P]),
    "recnn_hash_mask_dump": (_I, [_U, C.c_int32, _U, _I, _I, _P, _P]),
    "recnn_hash_mask_dump_at": (_I, [_U, _P, _I, _U, _I, _I, _P, _P]),
    "recnn_soft_update_flat": (_I, [_P, _P, _L, _F, _P]),
    "recnn_adam_flat": (_I, [_P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _I, _F, _P]),
    "recnn_adam_flat_shadow": (_I, [_P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _I, _F, _P, _P]),
    "recnn_adam_flat_at": (_I, [_P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _P, _I, _F, _P]),
    "recnn_l1_norm_flat": (_I, [_P, _L, _P, _P, _P]),
    "recnn_engine_query": (_I, [C.POINTER(EngineConfig), C.POINTER(EngineSizes)]),
    "recnn_engine_create": (_I, [C.POINTER(EngineConfig), _P, C.POINTER(_P)]),
    "recnn_engine_destroy": (None, [_P]),
    "recnn_engine_bind_net": (_I, [_P, _I, _P, _P, _P, _P]),
    "recnn_engine_bind_batch": (_I, [_P, _P, _P, _P, _P]),
    "recnn_engine_bind_slow": (_I, [_P, _I, _P]),
    "recnn_categorical_rows": (_I, [_P, _L, _I, _I, _I, C.c_uint32, C.c_int32, _P, _P, _P, _P]),
    "recnn_logprob_bwd": (_I, [_P, _L, _I, _I, _P, _P, _P, _P, _L, _I, _P, _P, _P]),
    "recnn_softmax_bwd": (_I, [_P, _L, _I, _I, _P, _L, _P, _L, _P]),
    "recnn_onehot_rows": (_I, [_P, _I, _I, _P, _L, _P]),
    "recnn_colsum_rows": (_I, [_P, _L, _I, _I, _P, _P]),
    "recnn_shard_softmax_pass": (_I, [_P, _L, _I, _I, _I, _P, _P, _P, _P]),
    "recnn_shard_logprob_bwd": (_I, [_P, _L, _I, _I, _P, _P, _P, _L, _P]),
    "recnn_transpose_rows": (_I, [_P, _L, _I, _I, _P, _L, _I, _P]),
    "recnn_vae_latent_fwd": (_I, [_P, _L, _P, _L, _I, _I, _P, _L, _P, _L, _P]),
    "recnn_vae_latent_bwd": (_I, [_P, _L, _P, _L, _P, _L, _P, _L, _P, _L, _P, _L, _I, _I, _P, _L, _P]),
    "recnn_vae_loss_fwd": (_I, [_P, _L, _P, _L, _P, _L, _P, _L, _I, _I, _I, _F, _P, _P, _P]),
    "recnn_vae_loss_bwd": (_I, [_P, _L, _P, _L, _P, _L, _P, _L, _I, _I, _I, _P, _F, _P, _L, _P, _L, _P, _L, _P]),
    "recnn_csr_workspace_bytes": (_I, [_L, C.POINTER(_L)]),
    "recnn_csr_build": (_I, [_P, _P, _P, _P, _L, _P, _P, _I, _P, _P, _P, _P, _P, _P, C.POINTER(_L), _P, _L, _P]),
    "recnn_ranger_flat": (_I, [_P, _P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _F, _I, _F, _I, _F, _P]),
    "recnn_ranger_flat_shadow": (_I, [_P, _P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _F, _I, _F, _I, _F, _P, _P]),
    "recnn_engine_bind_external": (_I, [_P, _P, _P]),
    "recnn_engine_bind_sampler": (_I, [_P, C.POINTER(Sampler)]),
    "recnn_engine_profile": (_I, [_P, _I, _I, _I, _P, C.POINTER(_F), C.POINTER(C.c_double), C.POINTER(C.c_char_p), C.POINTER(_I)]),
    "recnn_engine_set_hyper": (_I, [_P, C.POINTER(Hyper)]),
    "recnn_engine_set_mask_mode": (_I, [_P, _I]),
    "recnn_engine_refresh": (_I, [_P, _I, _P]),
    "recnn_engine_set_counters": (_I, [_P, _I, _I, _I, _I]),
    "recnn_engine_step": (_I, [_P, _I, _I, _I, _P]),
    "recnn_engine_value_grads": (_I, [_P, _I, _I, _P]),
    "recnn_engine_value_apply": (_I, [_P, _I, _F, _P]),
    "recnn_engine_policy_grads": (_I, [_P, _I, _I, _P]),
    "recnn_engine_policy_apply": (_I, [_P, _I, _F, _P]),
    "recnn_engine_clip_policy_grads": (_I, [_P, _F, _P]),
    "recnn_engine_soft_update": (_I, [_P, _I, _I, _F, _P]),
    "recnn_engine_finish": (_I, [_P, _I, _I, _I, _P]),
    "recnn_engine_graph_build": (_I, [_P, _I, _P]),
    "recnn_engine_graph_run": (_I, [_P, _I, _I, _P]),
    "recnn_engine_graph_prepare": (_I, [_P, _I, _I, _P]),
    "recnn_engine_dp_graph_build": (_I, [_P, _I, _F, _I, _P]),
    "recnn_engine_dp_graph_launch": (_I, [_P, _I, _P]),
    "recnn_comm_create": (_I, [_I, _I, _L, C.POINTER(_P)]),
    "recnn_comm_handle_bytes": (_L, []),
    "recnn_comm_export": (_I, [_P, _P, _L]),
    "recnn_comm_connect": (_I, [_P, _P, _L]),
    "recnn_dp_allreduce_flat": (_I, [_P, _P, _L, _P]),
    "recnn_comm_status": (_I, [_P, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "recnn_comm_set_timeout_ms": (_I, [_P, _I]),
    "recnn_comm_clear_status": (_I, [_P]),
    "recnn_comm_destroy": (None, [_P]),
    "recnn_engine_set_comm": (_I, [_P, _P, _F]),
    "recnn_comm_create_ex": (_I, [_I, _I, _L, _I, C.POINTER(_P)]),
    "recnn_comm_set_workgroups": (_I, [_P, _I]),
    "recnn_engine_read_losses": (_I, [_P, _P, _P]),
    "recnn_topk_item_aux": (_I, [_P, _I, _I, _I, _P, _P]),
    "recnn_topk_workspace_bytes": (_I, [_I, _I, C.POINTER(_L)]),
    "recnn_topk_search": (_I, [_P, _L, _I, _P, _I, _I, _I, _P, _I, _P, _P, _P, _P]),
    "recnn_engine_buffer": (_P, [_P, C.c_char_p, C.POINTER(_L), C.POINTER(_L), C.POINTER(_L), C.POINTER(_I)]),
}

# private debug / test hooks (recnn_amd/csrc/recnn_hip_debug.h): exported, but not part of the public header
DEBUG_SIGNATURES = {
    "recnn_debug_mlp_fault": (None, [_I]),
    "recnn_debug_mlp_probe": (None, [_I]),
    "recnn_debug_mlp_trace": (None, [_P]),
    "recnn_debug_tail_trace": (None, [_P]),
    "recnn_debug_frozen_trace": (None, [_P]),
    "recnn_debug_l1_trace": (None, [_P]),
    "recnn_debug_x3_fwd": (None, [_I]),
    "recnn_debug_x3_ws_probe": (None, [_I]),
    "recnn_debug_wide_ws": (None, [_I]),
    "recnn_debug_ws_trace": (None, [_P]),
    "recnn_debug_dwadam_trace": (None, [_P]),
}

_lib = None


def load():
    """Load librecnn_hip.so (after torch, so that both share one HIP runtime)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise RecnnHipMissing(
            f"{LIB_PATH} not found: build it with `make -C recnn_amd/csrc` "
            "(or __graft_entry__.build()).  recnn_amd has no CPU or PyTorch fallback.")
    import torch  # noqa: F401  -- loads libamdhip64.so.7 first; ours must bind to the same runtime
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in list(SIGNATURES.items()) + list(DEBUG_SIGNATURES.items()):
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    for which, st in enumerate((GemmArgs, EngineConfig, Hyper, EngineSizes, Sampler, EngineTuning, ShadowOut)):
        if lib.recnn_abi_sizeof(which) != C.sizeof(st):
            raise RecnnHipError(f"ABI mismatch for {st.__name__}: C={lib.recnn_abi_sizeof(which)} py={C.sizeof(st)}")
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().recnn_last_error()
        raise RecnnHipError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")


def call(name: str, *args):
    """Call an int-returning entry point and raise on a non-zero code."""
    check(getattr(load(), name)(*args), name)


def ptr(t):
    """Device (or host) address of a torch tensor, None -> NULL."""
    return None if t is None else C.c_void_p(t.data_ptr())


def current_stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
