// head.h -- argument blocks of the critic-head / loss kernels (head.hip).
#pragma once
#include "common.h"

constexpr int HEAD_MAX_CRITIC = 2;
constexpr int HEAD_ROWS_PER_BLOCK = 16;

struct HeadArgs {
  int rows, H, tc_bf16;   // tc_bf16: RECNN_F32 / RECNN_BF16 / RECNN_BF16X3 (compute type of the h2 / dz2 rows; split rows: ld_h physical)
  int64_t ld_h;
  // TD target side (n_target = 0: none, 1: DDPG, 2: TD3 min of twins)
  int n_target;
  const void* th2[2];
  const float* tw3[2];
  const float* tb3[2];
  const float* tq_in[2];  // non-NULL: target critic values already computed (chained inside the MLP launch); th2/tw3/tb3 unused
  const float* reward;
  const float* done;
  float gamma, lo, hi;
  float* expected;  // y[rows]
  float* target_q;  // optional debug: target critic value before the TD formula
  // learning critics
  int n_critic;
  const void* ch2[2];
  const float* cw3[2];
  const float* cb3[2];
  float* q[2];
  float* delta[2];      // 2 (q - y) / rows
  float* loss_part[2];  // per block partial sums: (q-y)^2, or q in policy mode
  int policy_mode;
  // fused backward seed (same rows, same block): dz2 = delta * w3 * scale * [h2 > 0] and the partial sums of
  // dW3 / db3 / db2 over the block's rows.  do_bwd = 0 skips it.
  int do_bwd;
  int train;             // dropout active: scale = 2
  float delta_const;     // policy mode: d(loss)/dQ = -1/rows for every row
  void* dz2[2];          // tc [rows, ld_h]
  float* dw3_part[2];    // [nblk][H] or NULL (no parameter gradients needed)
  float* db2_part[2];    // [nblk][H]
  float* db3_part[2];    // [nblk]
};

struct LossFinalizeArgs {
  int n;
  const float* part[4];
  int n_part[4];
  float scale[4];
  const float* add_ptr[4];  // optional: out[c] += add_scale[c] * (*add_ptr[c])
  float add_scale[4];
  float* out;  // [4]
  float* host_out;    // optional pinned host mirror: [0..3] = out[0..3], [4] = the hand-off error word out[4] as it stands when this kernel runs
  int n_tick;
  int32_t* tick[6];
  int tick_inc[6];     // *tick[i] += tick_inc[i]
  int32_t* wrap_ptr;  // sampler cursor: *wrap_ptr = (*wrap_ptr + wrap_inc) % wrap_mod
  int wrap_mod, wrap_inc;
  float* ring;        // optional loss history ring [ring_mask + 1][4]: entry (old *tick[0] + tick_inc[0] - 1) & ring_mask
  int ring_mask;
  int pair[4];        // 1: part[c] holds 2 n_part[c] half-panel sums (mlpt.hip, 16-row panels): partial i = part[2 i] + part[2 i + 1]
};

// Losses of the earlier steps of a run graph (their partial sums were kept per step; the run's last step is
// finalized by loss_finalize_kernel): one workgroup per step, written into the same history ring.
constexpr int LOSS_HIST_MAX = 64;
struct LossHistoryArgs {
  int n_steps;                 // steps 0 .. n_steps-1 of the run
  int n;                       // losses per step (<= 4); index n-1 is the policy loss
  const float* part[4];        // step 0's partial sums
  int64_t stride[4];           // floats between consecutive steps
  int n_part[4];
  float scale[4];
  int pol_count[LOSS_HIST_MAX];          // per step: number of policy-loss partials ...
  unsigned char pol_add[LOSS_HIST_MAX];  // ... and whether -b3 still has to be added (GEMM-epilogue partial dots)
  const float* b3;
  const int32_t* step_ctr;     // device step counter, not yet ticked for this run
  float* ring;
  int ring_mask;
  unsigned long long pair_steps;   // bit j: step j's value-loss partials are half-panel sums (LossFinalizeArgs.pair)
};
int loss_history_launch(const LossHistoryArgs& a, hipStream_t s);

struct GatherArgs;
// pregather != NULL: the sampler + gather of the NEXT step runs as extra workgroups of this launch
int head_launch(const HeadArgs& a, hipStream_t s, const GatherArgs* pregather = nullptr);
int loss_finalize_launch(const LossFinalizeArgs& a, hipStream_t s);
