// bwd.h -- argument block of the fused critic-head + layer-2 backward row-panel kernel (bwd.hip).
#pragma once
#include "common.h"

constexpr int BWD_MAX_GROUP = 2;
constexpr int BWD_ROWS = 32;  // rows per workgroup; also the granularity of every partial-sum slab it writes

struct BwdPanelProb {
  int rows, H;
  // ---- per-row seed d = dLoss/dQ
  int mode;                 // 0: TD error of a learning critic, 1: constant (policy loss through the critic)
  const float* q;           // mode 0: Q(s, a) of this critic, fp32 [rows] (written by the fused forward)
  int n_target;             // mode 0: 1 (DDPG) or 2 (TD3: min of the twins)
  const float* tq[2];       // target critic values, fp32 [rows]
  const float* reward;
  const float* done;
  float gamma, lo, hi;
  float delta_const;        // mode 1
  float* expected;          // mode 0 outputs (any may be NULL)
  float* target_q;
  float* delta_out;
  float* loss_part;         // [panels]: sum of (q - y)^2 over the panel's rows
  // ---- layer 3 backward: dz2 = d * w3 * scale * [h2 > 0]
  const void* h2;           // bf16 [rows, ldh]
  int64_t ldh;
  const float* w3;          // canonical fp32 [H]
  float scale;              // 2 when dropout is active (p = 0.5), else 1
  void* dz2;                // out bf16 [rows, ldh]
  float* dw3_part;          // [panels][H]   sum_r d_r h2[r, :]      (NULL: no parameter gradients wanted)
  float* db2_part;          // [panels][H]   sum_r dz2[r, :]
  float* db3_part;          // [panels]      sum_r d_r
  // ---- layer 2 backward: dz1 = (dz2 W2) * scale * [h1 > 0]
  const void* W2;           // bf16 shadow [H rows (out), ldw2] (in contiguous)
  int64_t ldw2;
  const void* h1;           // bf16 [rows, ldh]
  void* dz1;                // out bf16 [rows, ldh]
  float* colsum;            // db1 partial [panels][H] or NULL
};

struct BwdPanelBatch {
  BwdPanelProb p[BWD_MAX_GROUP];
};

// The whole gradient chain from the policy loss back into the actor for one 32-row panel (policy steps):
//   dz_e2 = d * w3c * s * [e2 > 0]          d = -1/B           (critic layer 3, ddpg.py:79)
//   dz_e1 = (dz_e2 W2c) * s * [e1 > 0]                          (critic layer 2)
//   dact  =  dz_e1 W1c[:, action columns]        -> db3 partial (critic layer 1 into the action = actor layer 3 output)
//   dz_p2 = (dact W3a) * s * [p2 > 0]            -> db2 partial (actor layer 3 -> 2)
//   dz_p1 = (dz_p2 W2a) * s * [p1 > 0]           -> db1 partial (actor layer 2 -> 1)
// dact, dz_p2, dz_p1 go to global memory (A operands of the actor's dW GEMMs); the critic's dz stay on chip.
struct BwdChainArgs {
  int rows, H, A;
  float delta_const, scale;
  int64_t ldh;              // pitch of the [rows, 256] bf16 activation / dz buffers
  const void* e2;           // policy-critic h2, h1 (bf16)
  const void* e1;
  const float* w3c;         // critic last layer, fp32 [H]
  const void* W2c; int64_t ldw2c;     // bf16 shadows, rows = out index
  const void* W1c; int64_t ldw1c;     // action columns are columns 0 .. A-1
  const void* W3a; int64_t ldw3a;     // [A rows, ld]
  const void* W2a; int64_t ldw2a;
  const void* p2;           // actor h2, h1
  const void* p1;
  void* dact; int64_t ldact;          // bf16 [rows, ldact]
  void* dzp2;
  void* dzp1;
  float* db3_part;          // [panels][A]
  float* db2_part;          // [panels][H]
  float* db1_part;          // [panels][H]
};
int bwd_chain_launch(const BwdChainArgs& a, hipStream_t s);

int bwd_init();
int bwd_panel_launch(const BwdPanelBatch& b, int nprob, hipStream_t s);
