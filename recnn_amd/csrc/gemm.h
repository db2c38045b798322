// gemm.h -- grouped MFMA GEMM descriptors shared by gemm.hip and engine.hip.
#pragma once
#include "common.h"

enum { GEMM_FWD = 0, GEMM_DX = 1, GEMM_DW = 2 };

struct GemmSeg {
  const void* A;
  const void* B;
  int64_t lda, ldb;
  int K;
};

// One GEMM problem.  Passed BY VALUE inside GemmBatch (kernel argument): no device-side
// descriptor memory, and hipGraph capture freezes it with the launch.
struct GemmProb {
  GemmSeg seg[2];
  int nseg;
  int M, N;
  void* C;
  int64_t ldc;
  int c_f32;
  // forward epilogue
  const float* bias;
  int relu;
  int mask_mode;
  const uint8_t* mask;
  int64_t ld_mask;
  uint32_t seed, stream_id;
  const int32_t* step_ptr;
  int step_add;            // mask key step = *step_ptr + step_add (steps captured ahead of the device counter)
  const float* addend;
  int64_t ld_add;
  float add_clip;
  int add_row_div;         // addend row of output row m is m / add_row_div (<= 1: m)
  // forward epilogue, optional: partial sums of sum_{m,n} C[m][n] * dot_w[n] (C as stored, i.e. rounded to the output
  // type), one per wave: dot_part[workgroup * waves + wave]; the launcher sets dot_parts to their number
  const float* dot_w;
  const float* dot_bias;   // optional scalar added once per output row (the N = 1 layer's bias, read at launch time)
  float* dot_part;
  int dot_parts;
  // dX epilogue
  const void* yref;
  int64_t ldy;
  float dx_scale;
  float* colsum;
  // dW: optional per-row (= per batch row = k) scale of the A operand: A[k][m] *= a_row_scale[k]  (unit backward
  // tensors times the per-row loss seed, mlp.h MlpCriticBwd)
  const float* a_row_scale;
  // dW epilogue
  int dw_splits;
  int64_t dw_slab_stride;
  int dw_valid_cols;
  int dw_col_rot;
  // filled by the launcher
  int tiles_m, tiles_n;
};

constexpr int GEMM_MAX_GROUP = 8;   // (TD3's split-bf16 layer-1 launch: 6 networks / parts + the deferred policy-loss critic)
struct GemmBatch {
  GemmProb p[GEMM_MAX_GROUP];
};

// Row-vector partial sums that ride on a dW launch as extra workgroups (one per 32-row panel and critic):
//   dw3_part[panel][k] = sum_r d_r h2[r][k],  db2_part[panel][k] = sum_r d_r u2[r][k],  colsum[panel][k] = sum_r d_r U[r][k]
struct DwVecProb {
  int rows, H;
  const float* delta;       // d_r, fp32 [rows]
  const void* h2;           // bf16 [rows, ldh]
  const void* u2;           // bf16 unit dz2
  const void* U;            // bf16 unit dz1
  int64_t ldh;
  float* dw3_part;          // [panels][H]
  float* db2_part;
  float* colsum;
};
struct DwVec {
  int n;
  DwVecProb p[2];
};

// Kernel / tile selection of the launchers (no process-wide state: an engine passes its own, the single-GEMM entry points the
// defaults).  All choices compute the same numbers.
struct GemmTune {
  int variant = -1;      // register-staged kernel tile: -1 = per launch, 0 = 64 x 64, 1 = 32 x 64 with a 2x longer k stage
  int v0_min_wg = 512;   // the per-launch heuristic takes the 64 x 64 tile from this many tiles on
  int dma = 1;           // forward GEMMs on compute-type operands use the LDS-DMA ring kernel
  int dma_deep = 1;      // 5-stage ring for launches of <= 320 tiles, else 3
  int dma_waves = 8;     // waves per workgroup of the LDS-DMA forward kernel (8 | 4)
  int waves = 8;         // waves per workgroup of the register-staged dX kernel (8 | 4)
  int x3_fwd = 2;        // split-bf16 forward GEMM: 2 = wave-specialised (loader waves + consumer waves), 11 = ... only for 64 x 128-tile
                         // launches, 0 = every wave loads and multiplies (round 4); gemm.hip x3_fwd_launch
  int dw_dma = 2;        // bf16 dW: 0 = register-staged; 1..7 = (rows per stage, ring slots) = (128,2) (64,2) (64,3) (64,4) (32,2) (32,4) (32,3)
};

// All problems of one launch share (dtype, mode, a_f32, b_f32).
struct GemmLaunch {
  int dtype, mode, a_f32, b_f32, nprob;
  GemmBatch batch;
  const DwVec* vec;   // dW launches only (bf16 DMA kernel), may be NULL
  const GemmTune* tune;   // NULL = defaults
  float* ws;              // forward, optional: scratch for split-K partial products (recnn_gemm_args::ws)
  int64_t ws_bytes;
};

void gemm_prob_init(GemmProb* p);
int gemm_launch(GemmLaunch* L, hipStream_t stream);
int gemm_init();  // one-time kernel attribute setup (call outside stream capture)
int gemm_from_args(const recnn_gemm_args* a, int mode, GemmLaunch* L);
