// engine_internal.h -- the step engine's state (struct recnn_engine) and the internal interface between its translation units:
//   engine.hip        workspace layout, create / bind / hyper entry points, the eager step API, read-backs, debug views
//   engine_plan.hip   optimizer layouts, problem builders, tuning entry points, the launch plans of a step (ph_*), step_impl
//   engine_graph.hip  per-launch profile, hipGraph capture (run-graph family, made-to-order graphs, cycle segments), graph replay,
//                     data-parallel phase graphs
// Internal names live in namespace recnn_eng (external linkage, so both units see ONE definition); nothing here is part of the
// C ABI (include/recnn_hip.h).
#pragma once
#include <string>
#include <vector>

#include "gather.h"
#include "gemm.h"
#include "head.h"
#include "mlp.h"
#include "bwd.h"
#include "optim.h"
#include "dwadam.h"
#include "comm.h"
#include "split.h"
#include "x3.h"
#include "x3tail.h"

constexpr int LOSS_RING = 1024;   // loss history ring entries (power of two)

// n_sets > 1: consecutive batches of n elements each, batch j under the key of step + step_add + j
int noise_fill_launch(float* out, int64_t n, float stddev, uint32_t seed, const int32_t* step_ptr, int step_add, hipStream_t s, int n_sets = 1);
int rows_to_bf16_launch(const float* xs, const float* xn, bf16_t* hs, bf16_t* hn, int rows, int64_t ld, hipStream_t s);

static inline int64_t ru(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

namespace recnn_eng {

constexpr int W1 = 0, B1 = 1, W2 = 2, B2 = 3, W3 = 4, B3 = 5;
constexpr int SP_W1_MAX = 8, SP_W2 = 16, SP_W3 = 16;  // max batch splits of the dW GEMMs

struct Net {
  bool critic = false, bound = false;
  float *p = nullptr, *g = nullptr, *m = nullptr, *v = nullptr;  // canonical flat arenas (caller owned)
  float* slow = nullptr;                                          // Lookahead slow weights (Ranger), caller owned
  int in_dim = 0, out_dim = 0;
  int64_t off[6] = {0, 0, 0, 0, 0, 0};
  int64_t n_params = 0;
  // compute-type shadows (workspace)
  char* shadow = nullptr;            // base of this net's shadow arena
  int64_t sh_off[6] = {-1, -1, -1, -1, -1, -1};  // element offsets (weights only)
  int ld_w1 = 0, ld_w2 = 0, ld_w3 = 0;
  int64_t shadow_elems = 0;
  // gradient partial slabs (workspace, learning nets only)
  float* gp[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  float* l1part = nullptr;
  int32_t* t_ptr = nullptr;
  int n_rows_blk = 0;
};

struct Acts {  // tc activations of one network application, [Bc, Hp]
  char *h1 = nullptr, *h2 = nullptr;
};

}  // namespace recnn_eng
using namespace recnn_eng;

struct recnn_engine {
  recnn_engine_config cfg;
  recnn_hyper hy;
  // Hp / Ap / K1a / K1c / ldx are extents of compute-type rows in ELEMENTS: for the split-bf16 type (x3.h) twice the logical
  // padded extents Hl / Al (weight-shadow row counts stay logical); ldx32 = row stride of the caller's fp32 packed rows
  int S, A, H, Hp, Ap, K1a, K1c, ldx, Bc, esz;
  int Hl, Al, ldx32;
  bool bf16, td3;
  recnn_engine_tuning tune;   // schedule / tile choices of THIS engine (recnn_engine_set_tuning)
  GemmTune gtune;             // ... the part of it the GEMM launchers read (gemm.h)
  bool x3 = false;     // compute type RECNN_BF16X3: split-bf16 rows through the layer-by-layer launches (gemm.hip / x3.hip)
  bool twins = false;  // the compute type differs from the bound fp32 rows: the step reads twins of the packed rows (workspace)
  int n_critic;
  char* ws = nullptr;
  int64_t ws_bytes = 0;
  Net net[RECNN_NET_COUNT];
  // batch
  float *xs = nullptr, *xn = nullptr, *reward = nullptr, *done = nullptr;
  const uint8_t* ext_masks = nullptr;
  const float* ext_noise = nullptr;
  // workspace buffers
  Acts tp, tq[2], cv[2], pa, pc;           // target policy, target critics, critics, actor, policy-critic
  char *dzc2[2], *dzc1[2];                 // critic backward
  char *dze2, *dze1, *dag, *dzp2, *dzp1;   // policy backward chain
  char *xcs = nullptr, *xcn = nullptr;     // packed rows in the compute type: the bound fp32 rows, or bf16 twins
  char *xsh = nullptr, *xnh = nullptr;     // bf16 twins (workspace, bf16 mode only)
  // second batch buffer set (bf16 sampler mode): inside a run graph the gather of step t+1 rides on the optimizer
  // launch of step t and fills the set step t is not reading
  char *xsh2 = nullptr, *xnh2 = nullptr;
  float *reward2 = nullptr, *done2 = nullptr;
  float *reward0 = nullptr, *done0 = nullptr;  // the bound reward / done arrays (set 0)
  int cur_set = 0;
  const GatherArgs* pregather = nullptr;   // set while the critic's optimizer launch should carry the next gather
  const GatherArgs* head_gather = nullptr; // ... or (split bf16 with the fused dW + optimizer launch) the critic head launch; NULL once consumed
  // run graphs: the device counters (mask-key step, Adam steps, sampler cursor) are ticked ONCE, by the finalize of
  // the run's last step; step i of the run is captured with these offsets on top of them
  int run_off = 0;                         // steps of the run before this one
  int run_t_off[RECNN_NET_COUNT] = {0};    // optimizer steps of each network earlier in the run
  bool run_skip_finish = false;            // not the last step of a run: no finalize launch
  int run_tick[3] = {1, 1, 1};             // increments applied by the finalize: steps, critic steps, actor steps
  char* gen_action;                        // tc [Bc, Ap] of the current batch buffer set
  char *gen_action0 = nullptr, *gen_action2 = nullptr;
  // deferred policy-loss forward of the previous step: that step's packed state rows and actor output, its run offset / slot
  struct PendingPc { bool on = false; const char* xs = nullptr; const char* ga = nullptr; int run_off = 0; int slot = 0; } pending_pc;
  // ---- cycle mode (round 3, bf16 sampler engines): the batches of up to MSET_MAX consecutive steps live side by side
  // (batch j = rows j * rows .. of every array), gathered by ONE launch and pushed through the FROZEN networks (target actor,
  // target critics, actor: they only change at a policy step) by one set of cycle-batched launches; the per-step launches
  // then carry the learning critics only (capture_run)
  static constexpr int MSET_MAX = 16;
  char *m_xs = nullptr, *m_xn = nullptr;           // bf16 [MSET_MAX * Bc, ldx]      (the CURRENT one of two copies: while a cycle
  float *m_reward = nullptr, *m_done = nullptr;    // [MSET_MAX * Bc]                  steps on one, a side branch of the run graph
  char *m_xs_b[2] = {nullptr, nullptr}, *m_xn_b[2] = {nullptr, nullptr};            //  gathers the next cycle's batches into the other)
  float *m_reward_b[2] = {nullptr, nullptr}, *m_done_b[2] = {nullptr, nullptr};
  char* m_ga = nullptr;                            // actor outputs, bf16 [MSET_MAX * Bc, Ap]
  float* m_tq[2] = {nullptr, nullptr};             // Q'(s', pi'(s')) per target critic, fp32 [MSET_MAX * Bc]
  float* m_noise = nullptr;                        // TD3 target-action noise, fp32 [MSET_MAX * Bc, A]
  char *m_tp_h1 = nullptr, *m_tp_h2 = nullptr, *m_pa_h1 = nullptr, *m_pa_h2 = nullptr, *m_tq_h1[2] = {nullptr, nullptr};   // bf16 [MSET_MAX * Bc, Hp]
  Acts pa0;                                        // the single-batch buffers the pointers below return to
  float* tqv0[2] = {nullptr, nullptr};
  float* noise_buf;                        // fp32 [Bc, A]
  float *expected, *target_q, *q[2], *delta[2], *qpi;
  bool panel_bwd_done = false;             // this step's critic head + dX ran in the bwd.hip launch
  float* tc_part[2];                       // chained target critics: fp32 [Bc, 256] layer-1 state parts
  int32_t* tc_flag[2];                     // ... their per-panel completion flags
  float* tqv[2];
  float* q_slot[2];                        // Q(s, a) hand-off slots (critic workgroup -> head in the target actor's workgroup)
  bool unit_bwd = false;                   // this step's dzc2 / dzc1 hold UNIT backward tensors (to be scaled by delta)
  bool half_panels = false;                // this step's tail launch ran 16-row panels: the small tensors' panel sums and the value-loss
                                           // partials are HALF-panel sums, consumed in pairs (TensorSeg.pair, LossFinalizeArgs.pair)
  bool hist_half[LOSS_HIST_MAX] = {};      // ... per step of the run being captured (loss history)
  float* pl_part;                          // policy loss: per-wave partial dots of the policy-critic's layer-2 GEMM
  int pl_cap = 0, pl_dot_parts = 0;        // capacity / number written by this step (0: the head kernel produced the loss)                           // ... their outputs, fp32 [Bc]
  float *loss_part[3];                     // value1, value2, policy  (per head block): the CURRENT step's slot of ...
  float *loss_part_base[3];                // ... LOSS_HIST_MAX per-step slots (run graphs keep every step's partial sums)
  int64_t loss_part_stride = 0;
  float* pl_part_base = nullptr;
  float* loss_ring = nullptr;              // [LOSS_RING][4] losses of the last LOSS_RING steps, indexed by the device step counter
  int hist_pol_count[LOSS_HIST_MAX];       // capture-time description of the run being captured
  unsigned char hist_pol_add[LOSS_HIST_MAX];
  float* losses;                           // device float[4]
  float* coef_out;                         // device float[1]
  int32_t* counters;                       // device int32[8]: step, t_policy, t_value1, t_value2
  float* l1_scratch;
  // step scalars of the optimizers (bias corrections ...: fp64 chains, optim.h) for every step of the run being issued:
  // [RUN_MAX][3] = {policy, value1, value2}; filled by one small launch at the start of a run graph / an eager step
  float* h_stage = nullptr;                // pinned, host-coherent: losses[0..3] + the hand-off error word, written by loss_finalize_kernel
  recnn_comm* comm = nullptr;              // data parallel: the gradient arenas are all-reduced in-stream (comm.hip)
  float comm_scale = 1.0f;                 // 1 / world
  bool comm_region = false;                // the arenas have regions of their own inside the communicator's buffers:
  int64_t comm_off[RECNN_NET_COUNT] = {};  //   gradients are produced into in[] and the optimizers read out[] (no copies)
  // device-resident sampler (optional)
  recnn_sampler smp;
  bool has_sampler = false;
  // per-launch profiler (recnn_engine_profile)
  bool prof_on = false;
  int prof_n = 0;
  int prof_repeat = 1;   // idempotent launches are issued this many times inside their event pair
  static constexpr int PROF_MAX = 48;
  hipEvent_t prof_ev[2 * PROF_MAX];
  const char* prof_name[PROF_MAX];
  double prof_flops[PROF_MAX];
  int prof_reps[PROF_MAX];
  bool prof_ready = false;
  // graphs
  // gexec[0] / gexec[1]: one ordinary step / one policy step.  Run graphs (several steps per graph launch):
  //   grun_o[k]  k ordinary steps                      (k >= 2; k = 1 is gexec[0])
  //   grun_p[k]  a policy step + k ordinary steps      (k >= 1; k = 0 is gexec[1]); k = policy_every-1 is a whole cycle
  //   grun_multi whole policy cycles, grun_multi_len steps (starts on a policy step)
  // graph_run() covers any (first_step, n_steps) with them: see recnn_engine_graph_run
  hipGraphExec_t gexec[2] = {nullptr, nullptr};
  static constexpr int RUN_MAX = 64;       // = LOSS_HIST_MAX: steps per run graph
  hipGraphExec_t grun_o[RUN_MAX + 1] = {};
  hipGraphExec_t grun_p[RUN_MAX + 1] = {};
  hipGraphExec_t grun_multi = nullptr;
  int grun_multi_len = 0;
  // run graphs made to order (recnn_engine_graph_prepare): one launch for a whole request (phase, n_steps)
  static constexpr int CUSTOM_MAX = 8;
  hipGraphExec_t grun_custom[CUSTOM_MAX] = {};
  int grun_custom_phase[CUSTOM_MAX] = {}, grun_custom_len[CUSTOM_MAX] = {};
  int grun_custom_next = 0;
  bool grun_look = false;    // run graphs alternate the two batch buffer sets (look-ahead gather)
  bool use_sampler = false;  // the step being issued / captured draws its batch from the bound sampler
  bool sampler_eager = false;  // eager public calls (recnn_engine_step, value_grads) draw from the sampler too
  hipGraphExec_t gdp[7][2] = {};           // data-parallel phase graphs [kind][batch buffer set]
  int dp_sets = 1;                         // 2: merged tail+head graphs alternate the batch buffer sets (look-ahead gather)
  int graph_rows = 0;
  bool hyper_set = false;
};

inline void sync_gemm_tune(recnn_engine* e) {
  const recnn_engine_tuning& t = e->tune;
  GemmTune& g = e->gtune;
  g.variant = t.gemm_variant; g.v0_min_wg = t.gemm_v0_threshold; g.dma = t.gemm_dma; g.dma_deep = t.gemm_dma_depth;
  g.dma_waves = t.gemm_dma_waves == 8 ? 8 : 4; g.waves = t.gemm_waves == 4 ? 4 : 8; g.dw_dma = t.dw_dma; g.x3_fwd = t.x3_fwd;
}

// Every kernel launch of the step goes through slot(): a no-op wrapper normally, a hipEvent pair in profile mode.
template <class F> int slot(recnn_engine* e, const char* name, double flops, hipStream_t s, F&& launch, bool idempotent = true) {
  if (!e->prof_on) return launch();
  const int i = e->prof_n;
  if (i >= recnn_engine::PROF_MAX) return launch();
  e->prof_name[i] = name;
  e->prof_flops[i] = flops;
  const int reps = idempotent ? e->prof_repeat : 1;
  e->prof_reps[i] = reps;
  (void)hipEventRecord(e->prof_ev[2 * i], s);
  int rc = 0;
  for (int r = 0; r < reps && !rc; ++r) rc = launch();
  (void)hipEventRecord(e->prof_ev[2 * i + 1], s);
  e->prof_n = i + 1;
  return rc;
}

// ---- the internal interface (defined in engine.hip / engine_plan.hip)
namespace recnn_eng {
int check_ready(recnn_engine* e, int rows);
NetLayout make_layout(const recnn_engine* e, int ni, int rows);
int apply_net(recnn_engine* e, int ni, int rows, bool do_adam, int opt_idx, float grad_scale, bool clip, int target_ni, float tau, hipStream_t s,
              bool from_slabs = false);
void drop_graphs(recnn_engine* e);
bool net_used(const recnn_engine* e, int ni);
bool lookahead_ok(const recnn_engine* e);
bool value_chain_ok(const recnn_engine* e);
bool value_panel_ok(const recnn_engine* e);
bool cycle_ok(const recnn_engine* e, int rows);
void use_hist_slot(recnn_engine* e, int i);
void use_set(recnn_engine* e, int k);
void use_mset(recnn_engine* e, int j, int rows);
void leave_mset(recnn_engine* e);
void select_mbuf(recnn_engine* e, int b);
int stage_batch(recnn_engine* e, int rows, hipStream_t s);
int frame_gather_packed(recnn_engine* e, int rows, hipStream_t s);
GatherArgs gather_args(const recnn_engine* e, int rows, int set, int cursor_add);
int ph_gather_cycle(recnn_engine* e, int rows, int n, int run_off0, int b, hipStream_t s);
int ph_frozen_batched(recnn_engine* e, int rows, int n, int run_off0, hipStream_t s);
int ph_forward(recnn_engine* e, int rows, bool value_side, bool actor_side, bool value_bwd, hipStream_t s);
int ph_value_backward(recnn_engine* e, int rows, bool reduce, hipStream_t s, bool dx_only = false);
int ph_policy(recnn_engine* e, int rows, bool backward, bool with_l1, hipStream_t s, bool need_rows = true);
int ph_policy_l1(recnn_engine* e, hipStream_t s);
int ph_finish(recnn_engine* e, int rows, bool ticked_value, bool ticked_policy, hipStream_t s);
int value_apply(recnn_engine* e, bool soft, float grad_scale, hipStream_t s, int rows = 0);
bool dwadam_ok(const recnn_engine* e, int rows);
int ph_value_dwadam(recnn_engine* e, int rows, bool soft, hipStream_t s);
int policy_apply(recnn_engine* e, bool soft, float grad_scale, hipStream_t s, bool have_l1 = false);
int net_allreduce(recnn_engine* e, int ni, const char* name, hipStream_t s);
int step_impl(recnn_engine* e, int rows, bool learn, bool policy_step, hipStream_t s, bool pregathered = false,
              bool gather_next = false, bool defer_policy_fwd = false, bool frozen_done = false);
}  // namespace recnn_eng
