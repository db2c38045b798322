/* recnn_hip_debug.h -- PRIVATE debug / test hooks of librecnn_hip.so: in-kernel shader-clock traces, timing probes and fault
 * injection.  Not part of the public C ABI (include/recnn_hip.h): process-wide, not thread-safe, results of probe runs are
 * garbage by design.  Used by tests/test_gpu_engine.py::test_broken_handoff_is_reported and tools/{mlp,split}_trace.py. */
#ifndef RECNN_HIP_DEBUG_H
#define RECNN_HIP_DEBUG_H
#ifdef __cplusplus
extern "C" {
#endif
/* break an in-launch hand-off of the fused forward on purpose: 1 = producers do not raise their flag, 2 = critics do not fill the
 * Q slot; the bounded waits then report through recnn_engine_read_losses / read_counters (RECNN_E_STATE).  0 = off. */
void recnn_debug_mlp_fault(int mode);
/* timing experiments on the fused forward (csrc/mlps.hip): bit 0 = no MFMA work / fragment reads, bit 1 = no DMA */
void recnn_debug_mlp_probe(int bits);
/* shader-clock stamps of a kernel's phases: device uint64 [workgroup][32] (mlps_fwd_kernel) / [workgroup][16]; NULL = off */
void recnn_debug_mlp_trace(void* device_u64_wg32);
void recnn_debug_tail_trace(void* device_u64_wg16);
void recnn_debug_frozen_trace(void* device_u64_wg16);   /* mlp_frozen_kernel: read at launch (= capture) time */
void recnn_debug_l1_trace(void* device_u64_wg16);
/* kernel variant of the split-bf16 forward GEMM for the single-problem entry point recnn_gemm_fwd (engines carry their own:
 * recnn_engine_tuning::x3_fwd); -1 = off */
void recnn_debug_x3_fwd(int variant);
/* timing experiments on the wave-specialised split-bf16 forward GEMM (results garbage): bit 0 consumers idle, bit 1 no DMA, bit 2 fragment
 * reads without MFMAs, bit 3 MFMAs without fragment reads, bit 4 no epilogue, bit 5 exit at entry, bit 6 direct epilogue stores, bit 7 plain
 * tile stores, bit 8 no kernel-argument prefetch, bit 9 the unpipelined consumer loop, bit 10 the general epilogue on full tiles (bits 9 and 10
 * leave the results intact: A/B runs of the step, RECNN_X3_WS_PROBE) */
void recnn_debug_x3_ws_probe(int bits);
/* catalogue-wide bf16 forward products of more than 128 rows: 1 (default) = 256 x 128 tiles on the wave-specialised kernel, 0 = round 3's
 * 128 x 128 kernel (A/B runs; the results are bit-identical); 2..5 = other tile / ring shapes (tools/wide_fwd_probe.py) */
void recnn_debug_wide_ws(int on);
/* shader-clock stamps of the plain-bf16 wave-specialised kernel's epilogue: device uint64 [workgroup][8] (loop end, barrier, image written,
 * barrier, stores issued, stores acknowledged); NULL = off */
void recnn_debug_ws_trace(void* device_u64_wg8);
/* shader-clock stamps of dw_adam_kernel's tile workgroups (csrc/dwadam.hip): device uint64 [workgroup][8] (entry, state requested, first stage
 * landed, contraction done, partial tiles in LDS, optimizer arithmetic done, stores issued, stores acknowledged); NULL = off */
void recnn_debug_dwadam_trace(void* device_u64_wg8);
#ifdef __cplusplus
}
#endif
#endif
