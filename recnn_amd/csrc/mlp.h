// mlp.h -- argument blocks of the fused row-panel MLP forward kernel (mlp.hip).
#pragma once
#include "common.h"

constexpr int MLP_MAX_GROUP = 8;
constexpr int MLP_MAX_TAIL = 2;
// Per-row scalars travel between workgroups of one launch through slots accessed with relaxed agent-scope atomics:
// the VALUE is the flag (slots rest at this NaN pattern), so the hand-off needs no L2 write-back / invalidate.
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
};

// Layer-2 backward of a learning critic appended to its forward workgroup (learn steps), in UNIT form: the workgroup
// still holds h2, both W2 k-slabs and its layer-1 relu/dropout gate bits on chip, so it emits
//     u2 = w3 * scale * [h2 > 0]                 (= dz2 / d)
//     U  = (u2 W2) * scale * [h1 > 0]            (= dz1 / d)
// where d = dLoss/dQ is a per-row scalar that depends on the target critic (another workgroup).  Nothing here waits
// for it: the consumers of dz2 / dz1 (the dW GEMM and the bias / last-layer partial sums riding on its launch,
// gemm.hip) multiply the rows by d.  Q(s, a) itself goes to the workgroup that evaluates the head through a
// value-as-flag hand-off slot.
struct MlpCriticBwd {
  int enabled;
  float* q_slot;     // [rows] hand-off slot for Q(s, a) (rests at MLP_TQ_EMPTY) or NULL
  float scale;       // 2 when dropout is active
  void* dz2;         // out bf16 [rows, ldh]: u2
  void* dz1;         // out bf16 [rows, ldh]: U
};

// TD target, TD error and loss of the learning critic(s), evaluated by the target actor's workgroup right after its
// chained target critics (it owns Q' of the rows; Q(s, a) arrives through the critics' hand-off slots, which were
// filled ~8 us earlier): recnn/nn/update/misc.py:6-7,33-39, td3.py:83-93.
struct MlpHead {
  int n_critic;              // 0: no head here
  float* q_slot[2];          // per critic, [rows]
  const float* reward;
  const float* done;
  float gamma, lo, hi;
  float* expected;           // y [rows]
  float* target_q;           // min_t Q'_t [rows]
  float* delta_out[2];       // d = 2 (q - y) / rows, per critic
  float* loss_part[2];       // [panels] sum (q - y)^2
  float* db3_part[2];        // [panels] sum d  (NULL: not wanted)
  // Cycle mode (engine.hip): Q' of the rows was computed for the whole policy cycle beforehand (mlpf.hip), so no workgroup of
  // THIS launch owns it -- a learning critic's own workgroup evaluates the head for its rows from self_tq (min over n_target of
  // them), its Q(s, a) never leaves the workgroup.  Same arithmetic, same reduction tree: the same bits.
  int n_target;
  const float* self_tq[2];   // fp32 [rows] each; self_tq[0] != NULL selects this mode (n_critic then counts the critics as usual)
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
  int n_tail;        // > 0: this (target actor) problem carries MlpBatch::tail[0 .. n_tail) and MlpBatch::head
  int cbwd_idx;      // >= 0: this critic problem carries MlpBatch::cbwd[cbwd_idx]; -1: none
};

// Kernel argument (by value, < 4 KB): the problems plus ONE copy of the chained-critic / head / critic-backward
// descriptions (a launch has at most one problem that carries tails).
struct MlpBatch {
  MlpProb p[MLP_MAX_GROUP];
  MlpTail tail[MLP_MAX_TAIL];
  MlpHead head;
  MlpCriticBwd cbwd[2];
  // Cross-workgroup waits are bounded: a wait that runs out ORs its bit into *err (device word, may be NULL) before the
  // workgroup carries on with whatever is in memory; the host side turns a non-zero word into RECNN_E_STATE when the
  // step's losses / counters are read (recnn_engine_read_losses), so a broken hand-off never passes as a number.
  int32_t* err;
  int spin_limit;    // polls before giving up (0: the default, ~0.2 s)
  int fault;         // test hook (recnn_debug_mlp_fault): 1 = producers do not raise their flag, 2 = critics do not fill the Q slot
  int nprob;
};
constexpr int MLP_ERR_PART_TIMEOUT = 1;   // layer-1 part of a chained target critic never arrived
constexpr int MLP_ERR_Q_TIMEOUT = 2;      // Q(s, a) hand-off slot never filled

int mlp_init();
int mlp_launch(const MlpBatch& b, int nprob, hipStream_t s);
int mlp_waves();  // waves per workgroup of the variant mlp_launch will use (MlpCriticBwd needs 16)
