// head.h -- argument blocks of the critic-head / loss kernels (head.hip).
#pragma once
#include "common.h"

constexpr int HEAD_MAX_CRITIC = 2;
constexpr int HEAD_ROWS_PER_BLOCK = 16;

struct HeadArgs {
  int rows, H, tc_bf16;
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
  int n_tick;
  int32_t* tick[6];
  int tick_inc[6];     // *tick[i] += tick_inc[i]
  int32_t* wrap_ptr;  // sampler cursor: *wrap_ptr = (*wrap_ptr + wrap_inc) % wrap_mod
  int wrap_mod, wrap_inc;
};

int head_launch(const HeadArgs& a, hipStream_t s);
int loss_finalize_launch(const LossFinalizeArgs& a, hipStream_t s);
