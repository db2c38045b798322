// mlp.h -- argument blocks of the fused row-panel MLP forward kernel (mlp.hip).
#pragma once
#include "common.h"

constexpr int MLP_MAX_GROUP = 6;
constexpr int MLP_MAX_TAIL = 2;
// Q' travels from a tail to the critic workgroups of the same rows through per-row slots accessed with relaxed
// agent-scope atomics: the VALUE is the flag (slots rest at this NaN pattern), so the hand-off on the step's critical
// path needs no L2 write-back / invalidate (a release-acquire pair there cost ~8 us).
constexpr uint32_t MLP_TQ_EMPTY = 0x7FC0DEADu;

// A critic chained behind an actor panel inside the same launch (target critic on [next_action | next_state]):
// the state part of its layer-1 pre-activation does not depend on the actor, so another workgroup of the launch
// (a `part_out` problem) computes it while the actor panel runs; the actor's workgroup then adds the action part
// and finishes layers 2 and 3 on chip.  `flag[panel]` goes 0 -> 1 when the part is complete (release) and back to
// 0 when consumed, so the buffers need no per-launch reset.  Producers must precede consumers in the launch order.
struct MlpTail {
  const float* part;               // fp32 [rows, 256]
  int32_t* flag;                   // [panels]
  const void* W1a; int64_t ldw1;   // action columns of the critic's W1 shadow: bf16 [256 rows, 128 k]
  const void* W2; int64_t ldw2;
  const float* b1;
  const float* b2;
  const float* b3;
  const float* w3row;
  float* q;                        // out: fp32 [rows]
  int n_ready;                     // consumers (critics with MlpCriticBwd) that take q through a hand-off slot
  float* ready_slot[2];            // [rows] each
};

// Backward seed and layer-2 backward of a learning critic appended to its forward workgroup (learn steps): once the
// chained target critic(s) of the same rows have delivered Q' (`tq_flag`), the workgroup -- which still holds h2, both
// W2 k-slabs and its layer-1 relu/dropout gate bits on chip -- computes the TD error, dz2, the dW3/db2/db3 partials,
// dz1 = (dz2 W2) * gate and the db1 partial.  Same outputs as bwd.hip's kernel, without its launch.
struct MlpCriticBwd {
  int enabled;
  int n_target;
  float* tq_slot[MLP_MAX_TAIL];     // [rows] hand-off slots of THIS consumer: hold MLP_TQ_EMPTY until the tail stores Q'
  const float* reward;
  const float* done;
  float gamma, lo, hi;
  float* expected;
  float* target_q;
  float* delta_out;
  float* loss_part;                 // [panels]
  float scale;                      // 2 when dropout is active
  void* dz2;                        // bf16 [rows, ldh]
  void* dz1;
  float* dw3_part;                  // [panels][H] (NULL: no parameter gradients wanted)
  float* db2_part;
  float* db3_part;                  // [panels]
  float* colsum;                    // db1 partial [panels][H]
};

struct MlpProb {
  // layer-1 input: up to two k-contiguous bf16 segments accumulated into the same pre-activation
  const void* A[2];
  int64_t lda[2];
  int K[2];          // multiples of 128
  int w1_col[2];     // first W1-shadow column of each segment
  int nseg;
  const void* W1; int64_t ldw1;   // bf16 shadow [256 rows, ldw1]
  const void* W2; int64_t ldw2;   // bf16 shadow [256, 256]
  const void* W3; int64_t ldw3;   // bf16 shadow [128 rows, 256] (actor) or NULL (critic)
  const float* b1;
  const float* b2;
  const float* b3;
  const float* w3row;             // critic: canonical fp32 [H]
  int rows, H, out_dim;
  // dropout of the two hidden layers
  int mask_mode;
  const uint8_t* mask1;
  const uint8_t* mask2;
  int64_t ld_mask;
  uint32_t seed, stream1, stream2;
  const int32_t* step_ptr;
  int step_add;      // mask key step = *step_ptr + step_add
  // outputs
  void* h1; void* h2; int64_t ldh;      // bf16 [rows, ldh]; either may be NULL
  void* out; int64_t ldo;                // actor output, bf16 [rows, ldo]
  float* q;                              // critic output, fp32 [rows]
  const float* addend; int64_t ld_add; float add_clip;
  // producer mode (layer 1 only, W2 unused): raw fp32 pre-activation part [rows, 256] + completion flags
  float* part_out;
  int32_t* part_flag;
  // consumer side: critics chained behind this actor's output
  int n_tail;
  MlpTail tail[MLP_MAX_TAIL];
  MlpCriticBwd cbwd;
};

struct MlpBatch {
  MlpProb p[MLP_MAX_GROUP];
};

int mlp_init();
int mlp_launch(const MlpBatch& b, int nprob, hipStream_t s);
int mlp_waves();  // waves per workgroup of the variant mlp_launch will use (MlpCriticBwd needs 16)
