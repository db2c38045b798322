"""Environment-variable access to librecnn_hip's tuning knobs (A/B experiments from the shell: `RECNN_DW_DMA=3 python bench.py`).

Nothing here changes what is computed; every knob selects among schedules / tile shapes that produce the same numbers.
`apply_env_knobs()` is called once by bench.py and the tools before the first engine is created."""
import os

from . import _lib as L

# environment variable -> C ABI knob (include/recnn_hip.h documents each)
KNOBS = {
    "RECNN_GEMM_VARIANT": "recnn_tune_gemm_variant",
    "RECNN_GEMM_DMA": "recnn_tune_gemm_dma",
    "RECNN_FUSED_MLP": "recnn_tune_fused_mlp",
    "RECNN_SAMPLER_F32": "recnn_tune_sampler_f32_rows",
    "RECNN_V0_MIN_WG": "recnn_tune_gemm_v0_threshold",
    "RECNN_MLP_PROBE": "recnn_tune_mlp_probe",
    "RECNN_GEMM_WAVES": "recnn_tune_gemm_waves",
    "RECNN_DMA_WAVES": "recnn_tune_gemm_dma_waves",
    "RECNN_DMA_DEEP": "recnn_tune_gemm_dma_depth",
    "RECNN_DEFER_PC": "recnn_tune_defer_policy_fwd",
    "RECNN_PREGATHER": "recnn_tune_pregather",
    "RECNN_GRAPH_RUN": "recnn_tune_graph_run",
    "RECNN_POLICY_CHAIN": "recnn_tune_policy_chain",
    "RECNN_BWD_PANEL": "recnn_tune_bwd_panel",
    "RECNN_CHAIN_TC": "recnn_tune_chain_target_critic",
    "RECNN_DW_DMA": "recnn_tune_dw_dma",
    "RECNN_DW_SPLITS": "recnn_tune_dw_splits",
    "RECNN_DW_FUSE": "recnn_tune_dw_fuse",
    "RECNN_OPT_TABLE": "recnn_tune_opt_table",
    "RECNN_MLP_XCD": "recnn_tune_mlp_xcd",
    "RECNN_CYCLE_FUSED_CRITIC": "recnn_tune_cycle_fused_critic",
    "RECNN_COMM_MEMORY": "recnn_tune_comm_memory",
    "RECNN_COMM_FUSED": "recnn_tune_comm_fused",
    "RECNN_COMM_WORKGROUPS": "recnn_tune_comm_workgroups",
    "RECNN_CYCLE_MIN_SEG": "recnn_tune_cycle_min_seg",
    "RECNN_SPLIT_FWD": "recnn_tune_split_fwd",
    "RECNN_CYCLE_MIN_LEN": "recnn_tune_cycle_min_len",
    "RECNN_L1_BIG": "recnn_tune_l1_big",
    "RECNN_FROZEN_GEMM": "recnn_tune_frozen_gemm",
    "RECNN_FROZEN_FUSED": "recnn_tune_frozen_fused",
    "RECNN_CYCLE_FORK": "recnn_tune_cycle_fork",
    "RECNN_DW_PROBE": "recnn_tune_dw_probe",
    "RECNN_GEMM_TGF": "recnn_tune_gemm_ks_layout",
    "RECNN_LD_PAD": "recnn_tune_ld_pad",
    "RECNN_GATHER_ROWS": "recnn_tune_gather_rows",
}


def apply_env_knobs():
    """Apply every knob whose environment variable is set; returns {variable: value} of what was applied."""
    lib = L.load()
    done = {}
    for var, fn in KNOBS.items():
        v = os.environ.get(var)
        if v is None or v == "":
            continue
        getattr(lib, fn)(int(v))
        done[var] = int(v)
    return done
