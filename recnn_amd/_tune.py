"""Tuning of the step engines from Python and from the environment (A/B experiments from the shell: `RECNN_DW_DMA=3 python bench.py`).

Nothing here changes what is computed: every field of `recnn_engine_tuning` (include/recnn_hip.h) selects among schedules / tile
shapes that produce the same numbers.  The C ABI keeps NO process-wide tuning state: each `StepEngine` receives its own copy
(`recnn_engine_set_tuning`), made here from the library defaults, then the environment, then `set_default_tuning(...)` overrides
(tests that run a whole facade under another schedule), then the engine's own `set_tuning(...)` keywords."""
import os
import warnings

from . import _lib as L

# environment variable -> recnn_engine_tuning field
ENV_FIELDS = {
    "RECNN_FUSED_MLP": "fused_mlp", "RECNN_CHAIN_TC": "chain_target_critic", "RECNN_BWD_PANEL": "bwd_panel",
    "RECNN_POLICY_CHAIN": "policy_chain", "RECNN_SPLIT_FWD": "split_fwd", "RECNN_CYCLE_MIN_LEN": "cycle_min_len",
    "RECNN_CYCLE_MIN_SEG": "cycle_min_seg", "RECNN_FROZEN_FUSED": "frozen_fused", "RECNN_FROZEN_GEMM": "frozen_gemm",
    "RECNN_GRAPH_RUN": "graph_run", "RECNN_PREGATHER": "pregather", "RECNN_DEFER_PC": "defer_policy_fwd",
    "RECNN_SAMPLER_F32": "sampler_f32_rows", "RECNN_DW_SPLITS": "dw_splits", "RECNN_COMM_FUSED": "comm_fused", "RECNN_L1_BIG": "l1_big",
    "RECNN_GEMM_VARIANT": "gemm_variant", "RECNN_V0_MIN_WG": "gemm_v0_threshold", "RECNN_GEMM_DMA": "gemm_dma",
    "RECNN_DMA_DEEP": "gemm_dma_depth", "RECNN_DMA_WAVES": "gemm_dma_waves", "RECNN_GEMM_WAVES": "gemm_waves", "RECNN_DW_DMA": "dw_dma",
    "RECNN_X3_TAIL": "x3_tail", "RECNN_X3_FWD": "x3_fwd", "RECNN_DW_FUSE": "dw_fuse", "RECNN_TAIL_HALF": "tail_half", "RECNN_L1_WS": "l1_ws", "RECNN_FROZEN_HALF": "frozen_half",
}
# process-level settings of the peer communicators (shared by engines: not part of an engine's tuning) and debug hooks
# (RECNN_COMM_MEMORY / RECNN_COMM_WORKGROUPS are read by each PeerComm for ITSELF: recnn_amd/parallel.py)
ENV_CALLS = {"RECNN_MLP_PROBE": "recnn_debug_mlp_probe", "RECNN_X3_FWD_DEBUG": "recnn_debug_x3_fwd", "RECNN_X3_WS_PROBE": "recnn_debug_x3_ws_probe",
             "RECNN_WIDE_WS": "recnn_debug_wide_ws"}

# debug hooks that make kernels compute garbage (timing probes): variable -> the bits that leave results intact
RESULT_CORRUPTING = {"RECNN_MLP_PROBE": 0, "RECNN_X3_WS_PROBE": (1 << 8) | (1 << 9) | (1 << 10)}
OVERRIDES_ENGINE_TUNING = {"RECNN_X3_FWD_DEBUG": "x3_fwd"}

_overrides = {}


def set_default_tuning(**fields):
    """Overrides for every engine created AFTERWARDS (None removes one): `set_default_tuning(split_fwd=2)`."""
    for k, v in fields.items():
        if k not in L.TUNING_FIELDS:
            raise KeyError(k)
        if v is None:
            _overrides.pop(k, None)
        else:
            _overrides[k] = int(v)


def make_tuning(**fields) -> "L.EngineTuning":
    t = L.EngineTuning()
    L.load().recnn_engine_tuning_init(t)
    for var, f in ENV_FIELDS.items():
        v = os.environ.get(var)
        if v not in (None, ""):
            setattr(t, f, int(v))
    for k, v in {**_overrides, **fields}.items():
        if k not in L.TUNING_FIELDS:
            raise KeyError(k)
        setattr(t, k, int(v))
    return t


def apply_env_knobs():
    """Apply the process-level settings whose environment variable is set (communicator memory / workgroups, debug probes); the
    per-engine fields are read from the environment whenever an engine is created.  Returns {variable: value} of everything set."""
    lib = L.load()
    done = {}
    for var, fn in ENV_CALLS.items():
        v = os.environ.get(var)
        if v not in (None, ""):
            val = int(v)
            if var in RESULT_CORRUPTING:
                # timing probes whose results are garbage by design: never from a stray variable alone
                keep = RESULT_CORRUPTING[var]
                if (val & ~keep) and os.environ.get("RECNN_ALLOW_GARBAGE_PROBES") != "1":
                    warnings.warn(f"{var}={val}: bits {val & ~keep:#x} make the kernels compute GARBAGE (timing probes); ignored -- set "
                                  "RECNN_ALLOW_GARBAGE_PROBES=1 to apply them", RuntimeWarning, stacklevel=2)
                    val &= keep
                elif val & ~keep:
                    warnings.warn(f"{var}={val}: timing probe active, results of the probed kernels are GARBAGE", RuntimeWarning, stacklevel=2)
                if val == 0:
                    continue
            elif var in OVERRIDES_ENGINE_TUNING:
                warnings.warn(f"{var}={val}: process-wide debug override of every engine's `{OVERRIDES_ENGINE_TUNING[var]}` tuning", RuntimeWarning,
                              stacklevel=2)
            getattr(lib, fn)(val)
            done[var] = val
    for var in ENV_FIELDS:
        v = os.environ.get(var)
        if v not in (None, ""):
            done[var] = int(v)
    return done
