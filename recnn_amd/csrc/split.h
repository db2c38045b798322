// split.h -- the split forward of the Actor / Critic MLPs (round 3): layer 1 as a full-machine tiled GEMM (l1gemm.hip), layers
// 2 / 3, the TD head and the critic's layer-2 backward as a lean row-panel "tail" kernel (mlpt.hip).
//
// Why split (profiles/NOTES_r01_r05.md 5c): layer 1 is 84-87 % of a network's weight bytes.  The fused row-panel kernel (mlps.hip) streams ALL
// of them through every 32-row workgroup (1.0-1.2 MB per workgroup at ~45 B/clk per CU = the whole launch time), and at 2048
// rows only a quarter to a half of the CUs have a workgroup of a given network.  A tiled layer-1 GEMM cuts N as well (64 x 64
// tiles: 393 KB per workgroup, every CU busy), and what remains per 32-row panel is 128-192 KB of weights.  The frozen networks
// (target actor, target critics: they only change at the policy step's soft update -- recnn/nn/update/ddpg.py:89-100,
// td3.py:130-141 -- and the actor, whose optimizer only steps there too) are applied to ALL batches of a policy cycle at once
// (M = cycle x rows: 128 x 128 tiles), so the per-step launches carry the learning critics only.
#pragma once
#include "common.h"

// ---------------------------------------------------------------------------------------------- layer 1
// h1[rows, ldh] = dropout(relu(A[rows, K] W1[H, K]^T + b1)) in bf16; A = up to two k-contiguous bf16 segments accumulated in
// order (segment 0 first: the critics contract [state | action] in THAT order, like mlps.hip's chained target critic).
struct L1Prob {
  const void* A[2];
  int64_t lda[2];
  int K[2];            // multiples of 128
  int w1_col[2];       // first W1-shadow column of each segment
  int nseg;
  const void* W1; int64_t ldw1;   // bf16 shadow [256 rows, ldw1]
  const float* b1;
  int rows, H;
  // dropout (train-mode networks): counter-based hash or external keep-masks, as GemmProb / MlpProb
  int mask_mode;
  const uint8_t* mask;
  int64_t ld_mask;
  uint32_t seed, stream;
  const int32_t* step_ptr;
  int step_add;
  // rows_per_set > 0: the rows are consecutive BATCHES of rows_per_set rows (a multiple of 16) -- batch j is step
  // (*step_ptr + step_add + j) of the run: mask key step + j, mask row = row inside the batch
  int rows_per_set;
  void* h1; int64_t ldh;
  // The same tiled kernel also runs the LATER layers of the cycle-batched frozen networks as plain GEMMs (K = 256: h2 = act(h1
  // W2^T + b2), out = h2 W3^T + b3 + clip(noise)), whose per-panel tail launches were weight-stream startups for 32 rows each:
  int no_relu;                  // 1: linear output (an actor's last layer)
  int w_rows;                   // rows of the weight matrix = output columns to cover (0: 256, the padded hidden width)
  const float* addend;          // optional fp32 [rows, ld_add] added after the bias, clamped to +- add_clip (TD3 target noise)
  int64_t ld_add;
  float add_clip;
  // filled by the launcher
  int tiles_m, tiles_n;
};
constexpr int L1_MAX_GROUP = 4;
struct L1Batch { L1Prob p[L1_MAX_GROUP]; };

int l1gemm_init();
// big = 0: 64 x 64 tiles (per-step launches, M ~ 2048-8192); 1: 128 x 128 tiles (cycle-batched launches, M >= 16k)
int l1gemm_launch(L1Batch& b, int nprob, int big, hipStream_t s, int ws = 0);   // ws: the 64 x 64 tile with loader / consumer waves

// ---------------------------------------------------------------------------------------------- tail
enum { TAIL_ACTOR = 0, TAIL_CRITIC_Q = 1, TAIL_CRITIC_LEARN = 2 };

struct TailProb {
  int kind;
  const void* h1; int64_t ldh;    // bf16 [rows, ldh]: layer-1 activations (dropout applied)
  const void* W2; int64_t ldw2;   // bf16 shadow [256, ldw2]
  const void* W3; int64_t ldw3;   // actor: bf16 shadow [128 rows, ldw3]
  const float* b2;
  const float* b3;
  const float* w3row;             // critics: canonical fp32 [H]
  int rows, H, out_dim;
  // dropout of layer 2
  int mask_mode;
  const uint8_t* mask2;
  int64_t ld_mask;
  uint32_t seed, stream2;
  const int32_t* step_ptr;
  int step_add;
  int rows_per_set;
  // outputs
  void* h2;                       // optional bf16 [rows, ldh]
  void* out; int64_t ldo;         // actor: bf16 [rows, ldo] (+ clip(addend))
  const float* addend; int64_t ld_add; float add_clip;
  float* q;                       // critics: fp32 [rows] (b3 included)
  // TAIL_CRITIC_LEARN: TD head + layer-2 backward of this critic (recnn/nn/update/misc.py:6-7,33-39, td3.py:83-93)
  int n_target;                   // target critics whose Q' enter the TD target (min over them)
  const float* tq[2];             // fp32 [rows] each, computed earlier (frozen networks)
  const float* reward;
  const float* done;
  float gamma, lo, hi;
  float* expected;                // y (first critic only, may be NULL)
  float* target_q;                // min_t Q'_t (first critic only, may be NULL)
  float* delta_out;               // d = 2 (q - y) / rows
  float* loss_part;               // [panels] sum (q - y)^2
  float scale;                    // 2 when dropout is active
  void* dz2;                      // out bf16 [rows, ldh]: d * w3 * scale * [h2 > 0]      (ready for the dW GEMM: no per-row scale)
  void* dz1;                      // out bf16 [rows, ldh]: d * ((u2 W2) * scale * [h1 > 0])
  float* dw3_part;                // [panels][H] sum_r d_r h2[r][.]     (NULL: no parameter gradients wanted)
  float* db2_part;                // [panels][H] sum_r d_r u2[r][.]
  float* db1_part;                // [panels][H] sum_r d_r U[r][.]
  float* db3_part;                // [panels]    sum_r d_r
  int half_panels;                // TAIL_CRITIC_LEARN: 16-row workgroups; every *_part array (and loss_part) then holds one entry per 16 rows =
                                  // the sum of that half panel's two 8-row chunks, and the consumers add entries 2 i and 2 i + 1 first -- the
                                  // (c0 + c1) + (c2 + c3) of a 32-row panel, bit for bit (optim_dev.h slab_grads, head.hip loss kernels)
};
constexpr int TAIL_MAX_GROUP = 4;
struct TailBatch { TailProb p[TAIL_MAX_GROUP]; };

int mlpt_init();
int mlpt_launch(const TailBatch& b, int nprob, hipStream_t s);
// q[m] = h2[m, :] . w3 + b3 for bf16 h2 [rows, ldh]: the critic head of mlp_tail_kernel as its own launch (one wave per row,
// the same lane -> column map and reduction tree, so the same bits), behind a cycle-batched layer-2 GEMM
int qdot_launch(const void* h2, int64_t ldh, const float* w3row, const float* b3, int H, int rows, float* q, hipStream_t s);

// ---------------------------------------------------------------------------------------------- frozen networks, cycle-wide
// One whole network (all three layers) on 128-row panels of M = cycle x rows rows (mlpf.hip): the target actor, the target
// critics on its output, the actor between its optimizer steps.
struct FrozenProb {
  const void* A[2];               // layer-1 input: up to two k-contiguous bf16 segments, contracted in order
  int64_t lda[2];
  int K[2];                       // multiples of 64
  int w1_col[2];
  int nseg;
  const void* W1; int64_t ldw1;
  const void* W2; int64_t ldw2;
  const void* W3; int64_t ldw3;   // actor (NULL: critic)
  const float* b1;
  const float* b2;
  const float* b3;
  const float* w3row;             // critic
  int rows, H, out_dim;
  int mask_mode;                  // RECNN_MASK_NONE | RECNN_MASK_HASH
  uint32_t seed, stream1, stream2;
  const int32_t* step_ptr;
  int step_add;
  int rows_per_set;               // batches of this many rows (a multiple of 32): batch j uses mask step + j
  void* h1; void* h2; int64_t ldh;   // optional bf16 [rows, ldh] (the actor's activations, for the policy step's backward)
  void* out; int64_t ldo;         // actor output, bf16
  const float* addend; int64_t ld_add; float add_clip;
  float* q;                       // critic output, fp32 [rows] (b3 included)
};
constexpr int FROZEN_MAX_GROUP = 4;
struct FrozenBatch {
  FrozenProb p[FROZEN_MAX_GROUP];
  unsigned long long* trace;      // debug: [workgroup][16] shader-clock stamps (set by mlpf_launch from recnn_debug_frozen_trace)
};
int mlpf_init();
// panel_rows: rows per workgroup, 128 | 64 (the same numbers: the launches of short cycle segments take 64 while they fit one round)
int mlpf_launch(const FrozenBatch& b, int nprob, hipStream_t s, int panel_rows = 128);
