// engine_plan.hip -- the launch plans of a DDPG / TD3 step (split out of engine.hip in round 5): optimizer layouts, GEMM / MLP problem
// builders, the per-engine tuning entry points, the phases of a step (forward in its four schedules: fused row panels, split, cycle-batched
// frozen networks, split bf16; value backward; policy loss and its backward chain; finish), the batch buffer sets, and step_impl.
// recnn/nn/update/ddpg.py:58-104, td3.py:66-150, misc.py:25-44 are what the sequence of launches replaces; state and the interface to the
// other units are engine_internal.h.
#include "engine_internal.h"

// ------------------------------------------------------------------------------------ layouts
namespace recnn_eng {

bool value_panel_ok(const recnn_engine* e);

NetLayout make_layout(const recnn_engine* e, int ni, int rows) {
  const Net& n = e->net[ni];
  NetLayout L;
  memset(&L, 0, sizeof(L));
  const int H = e->H;
  const int tiles_m = (rows + 31) / 32;
  const int nblk_hb = (rows + HEAD_ROWS_PER_BLOCK - 1) / HEAD_ROWS_PER_BLOCK;
  const int rr[6] = {H, 1, H, 1, n.out_dim, 1};
  const int cc[6] = {n.in_dim, H, H, H, H, n.out_dim};
  int blk = 0;
  for (int i = 0; i < 6; ++i) {
    TensorSeg& t = L.t[i];
    t.p_off = n.off[i];
    t.rows = rr[i];
    t.cols = cc[i];
    t.sh_off = n.sh_off[i];
    t.sh_ld = i == W1 ? n.ld_w1 : (i == W2 ? n.ld_w2 : n.ld_w3);
    t.col_rot = (i == W1 && n.critic) ? e->A : 0;
    t.gpart = n.gp[i];
    t.blk0 = blk;
    const int64_t ne = (int64_t)t.rows * t.cols;
    blk += opt_blocks(ne);
    t.small = opt_is_small(ne);
    t.nslab = 1;
    t.slab_stride = 0;
  }
  L.nblk = blk;
  L.n_params = n.n_params;
  if (rows > 0 && n.gp[W1]) {
    auto sp = [&](int mx) { int s = rows / 128; if (s < 1) s = 1; return s > mx ? mx : s; };
    L.t[W1].nslab = sp(e->tune.dw_splits); L.t[W1].slab_stride = (int64_t)H * n.in_dim;
    L.t[W2].nslab = sp(SP_W2); L.t[W2].slab_stride = (int64_t)H * H;
    L.t[B1].nslab = tiles_m; L.t[B1].slab_stride = H;
    if (n.critic) {
      const bool half = e->half_panels && e->panel_bwd_done && !e->unit_bwd;   // mlpt.hip ran 16-row panels: half-panel sums, taken in pairs
      const int nhead = (value_panel_ok(e) || half) ? tiles_m : nblk_hb;  // bwd.hip emits one partial per 32 rows, head.hip per 16
      L.t[W3].nslab = nhead; L.t[W3].slab_stride = H;
      L.t[B2].nslab = nhead; L.t[B2].slab_stride = H;
      L.t[B3].nslab = nhead; L.t[B3].slab_stride = 1;
      if (half) L.t[B1].pair = L.t[W3].pair = L.t[B2].pair = L.t[B3].pair = 1;
    } else {
      L.t[W3].nslab = sp(SP_W3); L.t[W3].slab_stride = (int64_t)e->A * H;
      L.t[B2].nslab = tiles_m; L.t[B2].slab_stride = H;
      L.t[B3].nslab = tiles_m; L.t[B3].slab_stride = e->A;
    }
  }
  for (int i = 0; i < 6; ++i) {
    TensorSeg& t = L.t[i];
    const int64_t ne = (int64_t)t.rows * t.cols;
    t.vec4 = !t.small && ne % 4 == 0 && t.p_off % 4 == 0 && (t.nslab <= 1 || t.slab_stride % 4 == 0) &&
             (((uintptr_t)t.gpart) & 15) == 0;
  }
  return L;
}

// byte offset of logical column `col` (a multiple of 32 for split rows) inside a compute-type row
inline int64_t tc_off(const recnn_engine* e, int col) { return (int64_t)(e->x3 ? 2 * col : col) * e->esz; }
inline char* sh_ptr(const recnn_engine* e, int ni, int which) {
  const Net& n = e->net[ni];
  return n.shadow + n.sh_off[which] * e->esz;
}

// (the finished per-rank gradient always goes to the bound arena; a collective copies it into the peer buffer itself, with
// system-scope stores)
inline float* g_produce(const recnn_engine* e, int ni) { return e->net[ni].g; }
// where its optimizer reads the gradient: a critic's, in region mode, straight from the collective's out[] (system-scope loads,
// ApplyArgs.g_sys); everything else from the bound arena, where the collective launch delivers the sums
inline bool g_direct(const recnn_engine* e, int ni) { return e->comm && e->comm_region && e->net[ni].critic; }
inline float* g_consume(const recnn_engine* e, int ni) { return g_direct(e, ni) ? comm_out(e->comm, e->comm_off[ni]) : e->net[ni].g; }
int net_allreduce(recnn_engine* e, int ni, const char* name, hipStream_t s);

// The optimizer / soft-update description of network `ni` (shared by apply_kernel launches and the fused dW epilogue).
int fill_apply_args(recnn_engine* e, int ni, const NetLayout& L, bool do_adam, int opt_idx, float grad_scale, bool clip, int target_ni,
                    float tau, ApplyArgs* out) {
  Net& n = e->net[ni];
  ApplyArgs& a = *out;
  memset(&a, 0, sizeof(a));
  a.p = n.p; a.g = g_consume(e, ni); a.m = n.m; a.v = n.v;
  a.g_sys = g_direct(e, ni);
  a.shadow = n.shadow;
  a.tc_bf16 = e->cfg.dtype;
  a.do_adam = do_adam;
  a.t_ptr = n.t_ptr;
  a.t_add = e->run_t_off[ni];
  if (do_adam) {
    RECNN_REQUIRE(n.g && n.m && n.v && n.t_ptr, "apply: network %d has no optimizer state bound", ni);
    a.lr = e->hy.lr[opt_idx]; a.beta1 = e->hy.beta1[opt_idx]; a.beta2 = e->hy.beta2[opt_idx];
    a.eps = e->hy.eps[opt_idx]; a.weight_decay = e->hy.weight_decay[opt_idx];
    a.opt_kind = e->hy.opt_kind[opt_idx];
    if (a.opt_kind == RECNN_OPT_RANGER) {
      RECNN_REQUIRE(n.slow, "apply: network %d runs Ranger but has no Lookahead slow-weight arena bound (recnn_engine_bind_slow)", ni);
      a.slow = n.slow; a.la_alpha = e->hy.la_alpha[opt_idx]; a.la_k = e->hy.la_k[opt_idx]; a.nsma_thr = e->hy.nsma_threshold[opt_idx];
    }
  }
  a.grad_scale = grad_scale;
  a.g_out = n.g;
  a.l1part = clip ? n.l1part : nullptr;
  a.n_l1 = clip ? L.nblk : 0;
  a.coef_out = clip ? e->coef_out : nullptr;
  if (target_ni >= 0) {
    a.tgt_p = e->net[target_ni].p;
    a.tgt_shadow = e->net[target_ni].shadow;
    a.tau = tau;
  }
  return 0;
}

int apply_net(recnn_engine* e, int ni, int rows, bool do_adam, int opt_idx, float grad_scale, bool clip, int target_ni,
              float tau, hipStream_t s, bool from_slabs) {
  Net& n = e->net[ni];
  NetLayout L = make_layout(e, ni, rows);
  ApplyArgs a;
  int rc = fill_apply_args(e, ni, L, do_adam, opt_idx, grad_scale, clip, target_ni, tau, &a);
  if (rc) return rc;
  a.from_slabs = from_slabs && do_adam && rows > 0;
  if (e->comm && a.from_slabs) {   // data parallel: the exchange runs inside this launch (comm_fused_ok checked by the caller)
    if ((rc = comm_port(e->comm, e->comm_off[ni], &a.comm))) return rc;
    a.comm_nwg = L.nblk;
    a.g = n.g; a.g_sys = 0;
  }
  const GatherArgs* pg = (do_adam && ni == RECNN_NET_VALUE1) ? e->pregather : nullptr;
  return slot(e, do_adam ? (n.critic ? (pg ? "adam_critic+gather" : "adam_critic") : "adam_actor") : "shadow_refresh", 0, s,
              [&] { return apply_launch(L, a, s, pg); }, !do_adam && target_ni < 0);
}

// ---- GEMM problem builders ---------------------------------------------------------------
struct FwdSpec {
  int ni;               // network whose weights are used
  int layer;            // 1, 2, 3
  const void* A; int64_t lda; int a_f32; int K;         // segment 0 input
  const void* A2 = nullptr; int64_t lda2 = 0; int K2 = 0; int b2_col = 0;  // optional second segment (B column offset)
  int b_col = 0;        // column offset into the W shadow for segment 0
  void* C; int64_t ldc; int c_f32;
  int relu;
  int mask_idx;         // -1 = none
  const float* addend = nullptr; int64_t ld_add = 0; float add_clip = 0.f;
  const float* dot_w = nullptr; const float* dot_bias = nullptr; float* dot_part = nullptr;
};

double fill_fwd(const recnn_engine* e, const FwdSpec& f, int rows, GemmProb* p) {
  const Net& n = e->net[f.ni];
  gemm_prob_init(p);
  const int wi = f.layer == 1 ? W1 : (f.layer == 2 ? W2 : W3);
  const int bi = wi + 1;
  const int ldw = f.layer == 1 ? n.ld_w1 : (f.layer == 2 ? n.ld_w2 : n.ld_w3);
  char* w = sh_ptr(e, f.ni, wi);
  p->seg[0].A = f.A; p->seg[0].lda = f.lda; p->seg[0].K = f.K;
  p->seg[0].B = w + tc_off(e, f.b_col); p->seg[0].ldb = ldw;
  p->nseg = 1;
  if (f.A2) {
    p->seg[1].A = f.A2; p->seg[1].lda = f.lda2; p->seg[1].K = f.K2;
    p->seg[1].B = w + tc_off(e, f.b2_col); p->seg[1].ldb = ldw;
    p->nseg = 2;
  }
  p->M = rows;
  p->N = f.layer == 3 ? n.out_dim : e->H;
  p->C = f.C; p->ldc = f.ldc; p->c_f32 = f.c_f32;
  p->bias = n.p + n.off[bi];
  p->relu = f.relu;
  p->mask_mode = RECNN_MASK_NONE;
  if (f.mask_idx >= 0 && e->cfg.mask_mode != RECNN_MASK_NONE) {
    p->mask_mode = e->cfg.mask_mode;
    if (e->cfg.mask_mode == RECNN_MASK_EXTERNAL) {
      p->mask = e->ext_masks + (int64_t)f.mask_idx * e->cfg.max_rows * e->H;
      p->ld_mask = e->H;
    } else {
      p->seed = e->cfg.seed;
      p->stream_id = (uint32_t)f.mask_idx;
      p->step_ptr = e->counters; p->step_add = e->run_off;
    }
  }
  p->addend = f.addend; p->ld_add = f.ld_add; p->add_clip = f.add_clip;
  p->dot_w = f.dot_w; p->dot_bias = f.dot_bias; p->dot_part = f.dot_part;
  const double kreal = f.layer == 1 ? n.in_dim : e->H;
  return 2.0 * rows * p->N * kreal;
}

struct Group {
  GemmLaunch L;
  double flops = 0.0;
  recnn_engine* eng;
  Group(recnn_engine* e, int mode, int a_f32, int b_f32) : eng(e) {
    memset(&L, 0, sizeof(L));
    L.dtype = e->cfg.dtype; L.mode = mode;
    L.a_f32 = e->bf16 ? a_f32 : 0;
    L.b_f32 = e->bf16 ? b_f32 : 0;
    L.nprob = 0;
    L.tune = &e->gtune;
  }
  GemmProb* add() { return &L.batch.p[L.nprob++]; }
  GemmProb* add(double fl) { flops += fl; return &L.batch.p[L.nprob++]; }
  int run(hipStream_t s, const char* name = "gemm") {
    return slot(eng, name, flops, s, [&] { return gemm_launch(&L, s); });
  }
};

// dX problem: C[rows, N] = (A[rows, Kc] * Wshadow[Kc, N(+col0)]) * scale*[yref>0]
double fill_dx(const recnn_engine* e, GemmProb* p, int rows, const void* A, int64_t lda, int Kc, int ni, int which, int col0,
             int N, void* C, int64_t ldc, const void* yref, int64_t ldy, float* colsum) {
  const Net& n = e->net[ni];
  gemm_prob_init(p);
  const int ldw = which == W1 ? n.ld_w1 : (which == W2 ? n.ld_w2 : n.ld_w3);
  p->seg[0].A = A; p->seg[0].lda = lda; p->seg[0].K = Kc;
  p->seg[0].B = sh_ptr(e, ni, which) + tc_off(e, col0); p->seg[0].ldb = ldw;
  p->M = rows; p->N = N;
  p->C = C; p->ldc = ldc; p->c_f32 = 0;
  p->yref = yref; p->ldy = ldy;
  p->dx_scale = yref ? (e->cfg.mask_mode != RECNN_MASK_NONE ? 2.0f : 1.0f) : 1.0f;
  p->colsum = colsum;
  const double kreal = (which == W3) ? n.out_dim : e->H;
  return 2.0 * rows * N * kreal;
}

// dW problem: slabs[s][M, valid] = dZ[rows, M]^T * X[rows, N]
double fill_dw(const recnn_engine* e, GemmProb* p, int rows, const void* dz, int64_t ldz, int M, const void* X, int64_t ldx_,
             int valid_cols, int rot, float* slabs, int splits, int64_t slab_stride) {
  gemm_prob_init(p);
  p->seg[0].A = dz; p->seg[0].lda = ldz; p->seg[0].K = rows;
  p->seg[0].B = X; p->seg[0].ldb = ldx_;
  p->M = M; p->N = valid_cols;
  p->C = slabs; p->ldc = valid_cols; p->c_f32 = 1;
  p->dw_splits = splits; p->dw_slab_stride = slab_stride;
  p->dw_valid_cols = valid_cols; p->dw_col_rot = rot;
  return 2.0 * rows * M * valid_cols;
}

int check_ready(recnn_engine* e, int rows) {
  RECNN_REQUIRE(e, "engine: null");
  RECNN_REQUIRE(rows > 0 && rows <= e->cfg.max_rows, "engine: rows=%d outside [1, %d]", rows, e->cfg.max_rows);
  RECNN_REQUIRE(e->xs && e->xn, "engine: batch buffers not bound");
  RECNN_REQUIRE(e->hyper_set, "engine: hyper-parameters not set");
  for (int ni = 0; ni < RECNN_NET_COUNT; ++ni)
    if (net_used(e, ni)) RECNN_REQUIRE(e->net[ni].bound, "engine: network %d not bound", ni);
  if (e->cfg.mask_mode == RECNN_MASK_EXTERNAL) RECNN_REQUIRE(e->ext_masks, "engine: external masks not bound");
  return 0;
}

// ---- per-engine tuning (include/recnn_hip.h recnn_engine_tuning): every field selects among schedules / tile shapes that
// produce the same numbers.  What the measured-slower variants of earlier rounds taught is recorded in profiles/NOTES_r01_r05.md 5c, not kept in
// the binary: the optimizer in the dW launch's epilogue (26.7 vs 18.7 us), an XCD-affine workgroup map of the fused forward
// (-11 MB of HBM traffic, +6 us), the cycle gather on a side branch of the run graph (67.5 vs 61.5 us/step), the learning
// critic's step forward as one fused launch in cycle mode (67.4 vs 65.1 us/step), padded leading dimensions (no effect).
//   fused_mlp: 0 = never, 1 = groups of >= 3 networks (a single network only occupies 64 CUs and streams its whole W1 per
//     workgroup: the tiled kernels are faster there), 2 = every forward
//   split_fwd: 0 = the fused row-panel kernel (mlps.hip) everywhere; 1 (default) = run graphs of at least cycle_min_len steps
//     run in CYCLE MODE (capture_run: the batches of a policy cycle gathered at once, the frozen networks applied to all of
//     them by mlpf.hip, per-step launches = split forward of the learning critics: l1gemm.hip + mlpt.hip); 2 = split forward and
//     cycle mode everywhere.  Measured (round 3, DDPG 2048 rows): sustained 61.2-61.5 us/step in cycle mode vs 62.9-63.2 fused;
//     a 20-step graph (the driver's command) 74.7 vs 69.9 -- three partial cycles do not pay there.
//   bwd_panel: 0 = head + dX launches, 1 = row-panel launch (bwd.hip), 2 = inside the critic's forward workgroup (mlps.hip)
extern "C" void recnn_engine_tuning_init(recnn_engine_tuning* t) {
  if (!t) return;
  memset(t, 0, sizeof(*t));
  t->fused_mlp = 1; t->chain_target_critic = 1; t->bwd_panel = 2; t->policy_chain = 1;
  t->split_fwd = 1; t->cycle_min_len = 20; t->cycle_min_seg = 4; t->frozen_fused = 1; t->frozen_gemm = 1;
  t->graph_run = -1; t->pregather = 1; t->defer_policy_fwd = 1;
  t->sampler_f32_rows = 0; t->dw_splits = 8; t->comm_fused = 1; t->l1_big = 1;
  t->gemm_variant = -1; t->gemm_v0_threshold = 512; t->gemm_dma = 1; t->gemm_dma_depth = 1; t->gemm_dma_waves = 8;
  t->gemm_waves = 8; t->dw_dma = 2; t->x3_tail = 1; t->x3_fwd = 2; t->dw_fuse = 1; t->tail_half = 1; t->l1_ws = 1; t->frozen_half = 1;
}
extern "C" int recnn_engine_set_tuning(recnn_engine* e, const recnn_engine_tuning* t) {
  RECNN_REQUIRE(e && t, "set_tuning: null pointer");
  e->tune = *t;
  recnn_engine_tuning& u = e->tune;
  u.dw_splits = u.dw_splits < 1 ? 1 : (u.dw_splits > SP_W1_MAX ? SP_W1_MAX : u.dw_splits);
  u.cycle_min_len = u.cycle_min_len < 2 ? 2 : u.cycle_min_len;
  u.cycle_min_seg = u.cycle_min_seg < 1 ? 1 : u.cycle_min_seg;
  u.l1_big = u.l1_big == 2 ? 2 : 1;
  sync_gemm_tune(e);
  drop_graphs(e);
  return 0;
}
extern "C" int recnn_engine_get_tuning(recnn_engine* e, recnn_engine_tuning* t) {
  RECNN_REQUIRE(e && t, "get_tuning: null pointer");
  *t = e->tune;
  return 0;
}

// bf16 value side on the fully fused path: target critics chained inside the forward launch (Q and Q' arrive as
// per-row scalars), critic head + first backward GEMM in one row-panel launch (bwd.hip)
bool value_chain_ok(const recnn_engine* e) {
  return e->tune.fused_mlp && e->tune.chain_target_critic && e->bf16 && e->Hp == 256 && e->Ap == 128 && e->A == e->Ap;
}
bool value_panel_ok(const recnn_engine* e) { return value_chain_ok(e) && e->tune.bwd_panel; }

bool fused_mlp_ok(const recnn_engine* e, int nprob) {
  return e->tune.fused_mlp && e->bf16 && e->Hp == 256 && e->Ap == 128 && (nprob >= 3 || e->tune.fused_mlp >= 2);
}

struct MlpSpec {
  int ni;
  const void* A0; int64_t lda0; int K0; int col0;
  const void* A1 = nullptr; int64_t lda1 = 0; int K1 = 0; int col1 = 0;
  void* h1 = nullptr; void* h2 = nullptr;
  void* out = nullptr; int64_t ldo = 0;
  float* q = nullptr;  // critic: Q per row
  int mask_idx = -1;   // external mask index of the first hidden layer (second = +1); -1 = eval mode
  const float* addend = nullptr; int64_t ld_add = 0; float add_clip = 0.f;
};

double fill_mlp(const recnn_engine* e, const MlpSpec& f, int rows, MlpProb* p) {
  const Net& n = e->net[f.ni];
  memset(p, 0, sizeof(*p));
  p->A[0] = f.A0; p->lda[0] = f.lda0; p->K[0] = f.K0; p->w1_col[0] = f.col0;
  p->nseg = 1;
  if (f.A1) { p->A[1] = f.A1; p->lda[1] = f.lda1; p->K[1] = f.K1; p->w1_col[1] = f.col1; p->nseg = 2; }
  p->W1 = sh_ptr(e, f.ni, W1); p->ldw1 = n.ld_w1;
  p->W2 = sh_ptr(e, f.ni, W2); p->ldw2 = n.ld_w2;
  if (!n.critic) { p->W3 = sh_ptr(e, f.ni, W3); p->ldw3 = n.ld_w3; }
  p->b1 = n.p + n.off[B1]; p->b2 = n.p + n.off[B2]; p->b3 = n.p + n.off[B3];
  p->w3row = n.critic ? n.p + n.off[W3] : nullptr;
  p->rows = rows; p->H = e->H; p->out_dim = n.out_dim;
  p->mask_mode = RECNN_MASK_NONE;
  if (f.mask_idx >= 0 && e->cfg.mask_mode != RECNN_MASK_NONE) {
    p->mask_mode = e->cfg.mask_mode;
    if (e->cfg.mask_mode == RECNN_MASK_EXTERNAL) {
      p->mask1 = e->ext_masks + (int64_t)f.mask_idx * e->cfg.max_rows * e->H;
      p->mask2 = e->ext_masks + (int64_t)(f.mask_idx + 1) * e->cfg.max_rows * e->H;
      p->ld_mask = e->H;
    } else {
      p->seed = e->cfg.seed; p->stream1 = (uint32_t)f.mask_idx; p->stream2 = (uint32_t)f.mask_idx + 1;
      p->step_ptr = e->counters; p->step_add = e->run_off;
    }
  }
  p->h1 = f.h1; p->h2 = f.h2; p->ldh = e->Hp;
  p->out = f.out; p->ldo = f.ldo;
  p->q = f.q;
  p->cbwd_idx = -1;
  p->addend = f.addend; p->ld_add = f.ld_add; p->add_clip = f.add_clip;
  return 2.0 * rows * ((double)e->H * n.in_dim + (double)e->H * e->H + (double)n.out_dim * e->H);
}

// ------------------------------------------------------------------------------------ phases
// ---- split forward (split.h): problem builders
void fill_l1(const recnn_engine* e, L1Prob* p, int ni, int rows, const void* A0, int64_t lda0, int K0, int col0, void* h1, int mask_idx,
             int step_add) {
  const Net& n = e->net[ni];
  memset(p, 0, sizeof(*p));
  p->A[0] = A0; p->lda[0] = lda0; p->K[0] = K0; p->w1_col[0] = col0; p->nseg = 1;
  p->W1 = sh_ptr(e, ni, W1); p->ldw1 = n.ld_w1;
  p->b1 = n.p + n.off[B1];
  p->rows = rows; p->H = e->H;
  p->mask_mode = RECNN_MASK_NONE;
  if (mask_idx >= 0 && e->cfg.mask_mode != RECNN_MASK_NONE) {
    p->mask_mode = e->cfg.mask_mode;
    if (e->cfg.mask_mode == RECNN_MASK_EXTERNAL) {
      p->mask = e->ext_masks + (int64_t)mask_idx * e->cfg.max_rows * e->H; p->ld_mask = e->H;
    } else {
      p->seed = e->cfg.seed; p->stream = (uint32_t)mask_idx; p->step_ptr = e->counters; p->step_add = step_add;
    }
  }
  p->h1 = h1; p->ldh = e->Hp;
}
void l1_seg1(L1Prob* p, const void* A1, int64_t lda1, int K1, int col1) {
  p->A[1] = A1; p->lda[1] = lda1; p->K[1] = K1; p->w1_col[1] = col1; p->nseg = 2;
}
void fill_tail(const recnn_engine* e, TailProb* p, int kind, int ni, int rows, const void* h1, int mask2_idx, int step_add) {
  const Net& n = e->net[ni];
  memset(p, 0, sizeof(*p));
  p->kind = kind;
  p->h1 = h1; p->ldh = e->Hp;
  p->W2 = sh_ptr(e, ni, W2); p->ldw2 = n.ld_w2;
  if (!n.critic) { p->W3 = sh_ptr(e, ni, W3); p->ldw3 = n.ld_w3; }
  p->b2 = n.p + n.off[B2]; p->b3 = n.p + n.off[B3];
  p->w3row = n.critic ? n.p + n.off[W3] : nullptr;
  p->rows = rows; p->H = e->H; p->out_dim = n.out_dim;
  p->mask_mode = RECNN_MASK_NONE;
  if (mask2_idx >= 0 && e->cfg.mask_mode != RECNN_MASK_NONE) {
    p->mask_mode = e->cfg.mask_mode;
    if (e->cfg.mask_mode == RECNN_MASK_EXTERNAL) {
      p->mask2 = e->ext_masks + (int64_t)mask2_idx * e->cfg.max_rows * e->H; p->ld_mask = e->H;
    } else {
      p->seed = e->cfg.seed; p->stream2 = (uint32_t)mask2_idx; p->step_ptr = e->counters; p->step_add = step_add;
    }
  }
}

// The forward of one step, split (e->tune.split_fwd): frozen networks first (target actor -> target critics on its action; the
// actor), then the learning critics with the TD head and their layer-2 backward in their own workgroups.  Six launches here;
// inside run graphs the frozen half is hoisted out of the step and applied to a whole policy cycle at once (capture_run).
int ph_forward_split(recnn_engine* e, int rows, bool value_side, bool actor_side, bool value_bwd, hipStream_t s, bool frozen_done = false) {
  const int A = e->A, nc = e->n_critic;
  const int POL = RECNN_NET_POLICY, TPOL = RECNN_NET_TARGET_POLICY;
  const int VAL[2] = {RECNN_NET_VALUE1, RECNN_NET_VALUE2}, TVAL[2] = {RECNN_NET_TARGET_VALUE1, RECNN_NET_TARGET_VALUE2};
  const int actor_m1 = e->td3 ? 4 : 2;
  const int64_t aoff = tc_off(e, A);
  const double l1_fl_a = 2.0 * rows * (double)e->H * e->S, l1_fl_c = 2.0 * rows * (double)e->H * (e->S + A);
  const double t_fl_a = 2.0 * rows * ((double)e->H * e->H + (double)A * e->H), t_fl_c = 2.0 * rows * ((double)e->H * e->H + e->H);
  int rc = 0;
  if (!frozen_done && value_side && e->td3 && !e->ext_noise) {
    if ((rc = slot(e, "td3_noise", 0, s, [&] { return noise_fill_launch(e->noise_buf, (int64_t)rows * A, e->hy.noise_std, e->cfg.seed, e->counters, e->run_off, s); }))) return rc;
  }
  if (!frozen_done) {  // ---- frozen networks, layer 1: target actor on s', actor on s
    L1Batch lb;
    int np = 0;
    double fl = 0;
    if (value_side) { fill_l1(e, &lb.p[np++], TPOL, rows, e->xcn + aoff, e->ldx, e->K1a, 0, e->tp.h1, -1, e->run_off); fl += l1_fl_a; }
    if (actor_side) { fill_l1(e, &lb.p[np++], POL, rows, e->xcs + aoff, e->ldx, e->K1a, 0, e->pa.h1, actor_m1, e->run_off); fl += l1_fl_a; }
    if (np && (rc = slot(e, "l1_actors", fl, s, [&] { return l1gemm_launch(lb, np, 0, s, e->tune.l1_ws); }))) return rc;
    TailBatch tb;
    np = 0; fl = 0;
    if (value_side) {   // next_action into the action slot of the packed next rows (+ TD3's clipped noise, td3.py:74-78)
      TailProb* p = &tb.p[np++];
      fill_tail(e, p, TAIL_ACTOR, TPOL, rows, e->tp.h1, -1, e->run_off);
      p->out = e->xcn; p->ldo = e->ldx;
      if (e->td3) { p->addend = e->ext_noise ? e->ext_noise : e->noise_buf; p->ld_add = A; p->add_clip = e->hy.noise_clip; }
      fl += t_fl_a;
    }
    if (actor_side) {
      TailProb* p = &tb.p[np++];
      fill_tail(e, p, TAIL_ACTOR, POL, rows, e->pa.h1, actor_m1 + 1, e->run_off);
      p->h2 = e->pa.h2; p->out = e->gen_action; p->ldo = e->Ap;
      fl += t_fl_a;
    }
    if (np && (rc = slot(e, "tail_actors", fl, s, [&] { return mlpt_launch(tb, np, s); }))) return rc;
  }
  e->panel_bwd_done = false;
  e->unit_bwd = false;
  e->half_panels = false;
  if (!value_side) return 0;
  // 16-row panels for the learning critics' tail (tuning.tail_half): whole 32-row pairs only, and the consumers of its panel sums must
  // be the pair-aware ones (the optimizer launches; not grad_reduce's callers that read gp[] directly -- there are none)
  // ... and only while the launch still fits the machine in ONE round of workgroups (one 152 KB workgroup per CU): 2 nc (rows / 32) half panels
  // + rows / 32 panels of the deferred policy-loss critic <= 256.  DDPG at 2048 rows: 192 (53.9 vs 57.3 us/step); TD3 at 4096 rows would be
  // 640 -- measured 130.5 vs 124.1 us/step with 32-row panels (profiles/NOTES_r06.md)
  const bool half = e->tune.tail_half && value_bwd && rows % 32 == 0 && (2 * nc + 1) * (rows / 32) <= 256;
  if (!frozen_done) {  // ---- target critics on [next_state | next_action]: state part first, then the action columns (mlps.hip's chained order)
    L1Batch lb;
    TailBatch tb;
    double fl = 0, tfl = 0;
    for (int c = 0; c < nc; ++c) {
      fill_l1(e, &lb.p[c], TVAL[c], rows, e->xcn + aoff, e->ldx, e->K1a, A, e->tq[c].h1, -1, e->run_off);
      l1_seg1(&lb.p[c], e->xcn, e->ldx, e->Ap, 0);
      fill_tail(e, &tb.p[c], TAIL_CRITIC_Q, TVAL[c], rows, e->tq[c].h1, -1, e->run_off);
      tb.p[c].q = e->tqv[c];
      fl += l1_fl_c; tfl += t_fl_c;
    }
    if ((rc = slot(e, "l1_target_critic", fl, s, [&] { return l1gemm_launch(lb, nc, 0, s, e->tune.l1_ws); }))) return rc;
    if ((rc = slot(e, "tail_target_critic", tfl, s, [&] { return mlpt_launch(tb, nc, s); }))) return rc;
  }
  {  // ---- learning critics (+ the previous step's policy-loss forward riding along: same weights, the previous batch)
    L1Batch lb;
    TailBatch tb;
    int np = 0;
    double fl = 0, tfl = 0;
    const bool train = e->cfg.mask_mode != RECNN_MASK_NONE;
    for (int c = 0; c < nc; ++c) {
      Net& v = e->net[VAL[c]];
      fill_l1(e, &lb.p[np], VAL[c], rows, e->xcs, e->ldx, e->K1c, 0, e->cv[c].h1, 2 * c, e->run_off);
      TailProb* p = &tb.p[np++];
      fill_tail(e, p, TAIL_CRITIC_LEARN, VAL[c], rows, e->cv[c].h1, 2 * c + 1, e->run_off);
      p->h2 = e->cv[c].h2; p->q = e->q[c];
      p->n_target = nc;
      for (int t = 0; t < nc; ++t) p->tq[t] = e->tqv[t];
      p->reward = e->reward; p->done = e->done; p->gamma = e->hy.gamma;
      p->lo = e->td3 ? -INFINITY : e->hy.min_value;
      p->hi = e->td3 ? INFINITY : e->hy.max_value;
      if (c == 0) { p->expected = e->expected; p->target_q = e->target_q; }
      p->delta_out = e->delta[c]; p->loss_part = e->loss_part[c];
      p->scale = train ? 2.0f : 1.0f;
      p->half_panels = half;
      p->dz2 = e->dzc2[c]; p->dz1 = e->dzc1[c];
      if (value_bwd) {
        RECNN_REQUIRE(v.g, "value backward: network %d has no gradient arena bound", VAL[c]);
        p->dw3_part = v.gp[W3]; p->db2_part = v.gp[B2]; p->db1_part = v.gp[B1]; p->db3_part = v.gp[B3];
      }
      fl += l1_fl_c; tfl += t_fl_c + 2.0 * rows * (double)e->H * e->H;
    }
    if (e->pending_pc.on && np < L1_MAX_GROUP) {
      const auto& pp = e->pending_pc;
      const int m0 = e->td3 ? 6 : 4;
      fill_l1(e, &lb.p[np], RECNN_NET_VALUE1, rows, pp.ga, e->Ap, e->Ap, 0, e->pc.h1, m0, pp.run_off);
      l1_seg1(&lb.p[np], pp.xs + aoff, e->ldx, e->K1a, A);
      TailProb* p = &tb.p[np++];
      fill_tail(e, p, TAIL_CRITIC_Q, RECNN_NET_VALUE1, rows, e->pc.h1, m0 + 1, pp.run_off);
      p->q = e->pl_part_base + (int64_t)pp.slot * e->pl_cap;   // that step's policy-loss slot: Q per row, b3 included
      fl += l1_fl_c; tfl += t_fl_c;
      e->pending_pc.on = false;
    }
    // 64 x 64 tiles while they fit two rounds of workgroups; beyond that (TD3 at 4096 rows: 3 problems x 256 tiles = three rounds of 8 us)
    // the 128 x 128 form -- a quarter of the workgroups, twice the stream each: one round (122.35 -> 120.1 us/step, A/B in one call)
    const int tiles64 = np * ((rows + 63) / 64) * 4;
    const int big = tiles64 > 512 ? 1 : 0;
    if ((rc = slot(e, "l1_critic", fl, s, [&] { return l1gemm_launch(lb, np, big, s, e->tune.l1_ws); }))) return rc;
    if ((rc = slot(e, "tail_critic", tfl, s, [&] { return mlpt_launch(tb, np, s); }))) return rc;
  }
  e->panel_bwd_done = true;     // dz2 / dz1 (already times the per-row loss seed) and the small tensors' panel sums exist
  e->half_panels = half;
  return 0;
}

// layers 2 (+ 3) of network ni as a problem of the split-bf16 row-panel launch (x3tail.hip); mask_idx: dropout stream of layer 2
void fill_x3tail(const recnn_engine* e, X3TailProb* p, int ni, int rows, const void* h1, int mask_idx, int step_add) {
  const Net& n = e->net[ni];
  memset(p, 0, sizeof(*p));
  p->h1 = h1; p->ldh = e->Hp;
  p->W2 = sh_ptr(e, ni, W2); p->ldw2 = n.ld_w2;
  p->b2 = n.p + n.off[B2]; p->b3 = n.p + n.off[B3];
  if (n.out_dim > 1) { p->W3 = sh_ptr(e, ni, W3); p->ldw3 = n.ld_w3; }
  else p->w3row = n.p + n.off[W3];
  p->rows = rows; p->H = e->H; p->out_dim = n.out_dim;
  p->mask_mode = RECNN_MASK_NONE;
  if (mask_idx >= 0 && e->cfg.mask_mode != RECNN_MASK_NONE) {
    p->mask_mode = e->cfg.mask_mode;
    if (e->cfg.mask_mode == RECNN_MASK_EXTERNAL) { p->mask2 = e->ext_masks + (int64_t)mask_idx * e->cfg.max_rows * e->H; p->ld_mask = e->H; }
    else { p->seed = e->cfg.seed; p->stream2 = (uint32_t)mask_idx; p->step_ptr = e->counters; p->step_add = step_add; }
  }
}
bool x3_tail_ok(const recnn_engine* e) {
  return e->x3 && e->tune.x3_tail && e->H == 256 && e->Hp == 512 && e->A == 128 && e->net[RECNN_NET_POLICY].ld_w2 == e->net[RECNN_NET_POLICY].ld_w3;
}


// The forward of one step in the split-bf16 type (x3.h): layer-by-layer GEMM launches like the fp32 path, arranged so that every
// launch is as full as the data dependencies allow (each launch streams its tiles at one CU's L2 -> LDS rate: time = bytes / CUs):
//   L1  {target actor(s'), critic(s, a), actor(s), target critic STATE part (raw fp32, 1290 of its 1418 k: it does not depend on
//        the target actor), [the previous step's policy-loss critic on [pi(s) | s]: deferred, run graphs]}
//   L2  {target actor, critic, actor, [deferred policy-loss critic, the loss summed in the epilogue]}
//   L3  {target actor -> next_action (+ TD3 noise), actor -> gen_action}
//   L1' target critic: relu(state part + next_action W1[:, action columns]^T + b1)      (k = 128)
//   L2' target critic        then the head launch (TD target, Q dots, loss partials, dz2).
// recnn/nn/update/misc.py:27-39, td3.py:73-93, ddpg.py:78-79 (deferred), same arithmetic as ph_forward's generic branch up to the
// summation order of the target critic's layer 1.
int ph_forward_x3(recnn_engine* e, int rows, bool value_side, bool actor_side, bool value_bwd, hipStream_t s) {
  const int A = e->A, Hp = e->Hp, nc = e->n_critic;
  const int POL = RECNN_NET_POLICY, TPOL = RECNN_NET_TARGET_POLICY;
  const int VAL[2] = {RECNN_NET_VALUE1, RECNN_NET_VALUE2}, TVAL[2] = {RECNN_NET_TARGET_VALUE1, RECNN_NET_TARGET_VALUE2};
  const int actor_m1 = e->td3 ? 4 : 2;
  const int64_t aoff = tc_off(e, A);
  const int ldp = e->Hl > 256 ? e->Hl : 256;     // row pitch of the fp32 layer-1 parts
  int rc;
  const bool pend = e->pending_pc.on;
  const recnn_engine::PendingPc pp = e->pending_pc;
  e->pending_pc.on = false;
  const int m0 = e->td3 ? 6 : 4;
  // DDPG's first launch is exactly one round of 64 x 128 tiles on 256 CUs without the deferred policy-loss critic (4 problems x 64
  // tiles); a fifth problem would cost a whole second round (measured 44.6 us against ~28).  It rides with the target critic's
  // k = 128 launch and panel tail instead, which fill a quarter of the machine.  (TD3's first launch is two rounds either way.)
  const bool pend_late = pend && value_side && !e->td3 && x3_tail_ok(e);
  auto pc_l1 = [&](Group& g) {     // layer 1 of the deferred policy-loss critic on [pi(s) | s] of the previous batch
    FwdSpec f{RECNN_NET_VALUE1, 1, pp.ga, e->Ap, 0, e->Ap};
    f.b_col = 0;
    f.A2 = pp.xs + aoff; f.lda2 = e->ldx; f.K2 = e->K1a; f.b2_col = A;
    f.C = e->pc.h1; f.ldc = Hp; f.c_f32 = 0; f.relu = 1; f.mask_idx = m0;
    GemmProb* p = g.add();
    g.flops += fill_fwd(e, f, rows, p);
    p->step_add = pp.run_off;
  };
  auto pc_tail = [&](X3TailProb* p) -> int {
    fill_x3tail(e, p, RECNN_NET_VALUE1, rows, e->pc.h1, m0 + 1, pp.run_off);
    p->q_part = e->pl_part_base + (int64_t)pp.slot * e->pl_cap;
    const int parts = ((rows + 31) / 32) * x3tail_parts_per_panel();
    RECNN_REQUIRE(parts <= e->pl_cap, "policy loss: %d partial sums do not fit %d", parts, e->pl_cap);
    if (pp.slot < LOSS_HIST_MAX) { e->hist_pol_count[pp.slot] = parts; e->hist_pol_add[pp.slot] = 0; }
    return 0;
  };
  {  // ---- L1
    Group g(e, GEMM_FWD, 0, 0);
    if (value_side) {
      FwdSpec f{TPOL, 1, e->xcn + aoff, e->ldx, 0, e->K1a};
      f.C = e->tp.h1; f.ldc = Hp; f.c_f32 = 0; f.relu = 1; f.mask_idx = -1;
      g.flops += fill_fwd(e, f, rows, g.add());
      for (int c = 0; c < nc; ++c) {
        FwdSpec fc{VAL[c], 1, e->xcs, e->ldx, 0, e->K1c};
        fc.C = e->cv[c].h1; fc.ldc = Hp; fc.c_f32 = 0; fc.relu = 1; fc.mask_idx = 2 * c;
        g.flops += fill_fwd(e, fc, rows, g.add());
      }
    }
    if (actor_side) {
      FwdSpec f{POL, 1, e->xcs + aoff, e->ldx, 0, e->K1a};
      f.C = e->pa.h1; f.ldc = Hp; f.c_f32 = 0; f.relu = 1; f.mask_idx = actor_m1;
      g.flops += fill_fwd(e, f, rows, g.add());
    }
    if (value_side) {
      for (int c = 0; c < nc; ++c) {   // state columns of the target critic's W1 shadow ([action | state] order): raw fp32, no bias
        FwdSpec f{TVAL[c], 1, e->xcn + aoff, e->ldx, 0, e->K1a};
        f.b_col = A;
        f.C = e->tc_part[c]; f.ldc = ldp; f.c_f32 = 1; f.relu = 0; f.mask_idx = -1;
        GemmProb* p = g.add();
        fill_fwd(e, f, rows, p);
        p->bias = nullptr;
        g.flops += 2.0 * rows * (double)e->H * e->S;
      }
    }
    if (pend && !pend_late && g.L.nprob < GEMM_MAX_GROUP) pc_l1(g);
    if ((rc = g.run(s, "fwd_l1"))) return rc;
  }
  const bool tails = x3_tail_ok(e);
  if (value_side && e->td3 && !e->ext_noise) {
    if ((rc = slot(e, "td3_noise", 0, s, [&] { return noise_fill_launch(e->noise_buf, (int64_t)rows * A, e->hy.noise_std, e->cfg.seed, e->counters, e->run_off, s); }))) return rc;
  }
  if (tails) {
    // ---- layers 2 + 3 of everything whose layer 1 exists, ONE launch of 32-row panels (x3tail.hip): target actor -> next_action
    // (+ TD3 noise), critics -> h2 (the head and the backward read it), actor -> h2, gen_action, deferred policy-loss critic -> sums of Q
    X3TailBatch tb;
    int np = 0;
    double fl = 0;
    const double fl2 = 2.0 * rows * (double)e->H * e->H;
    if (value_side) {
      X3TailProb* p = &tb.p[np++];
      fill_x3tail(e, p, TPOL, rows, e->tp.h1, -1, e->run_off);
      p->out = e->xcn; p->ldo = e->ldx;
      if (e->td3) { p->addend = e->ext_noise ? e->ext_noise : e->noise_buf; p->ld_add = A; p->add_clip = e->hy.noise_clip; }
      fl += fl2 + 2.0 * rows * (double)e->H * A;
      for (int c = 0; c < nc; ++c) {
        p = &tb.p[np++];
        fill_x3tail(e, p, VAL[c], rows, e->cv[c].h1, 2 * c + 1, e->run_off);
        p->h2 = e->cv[c].h2;
        fl += fl2;
      }
    }
    if (actor_side) {
      X3TailProb* p = &tb.p[np++];
      fill_x3tail(e, p, POL, rows, e->pa.h1, actor_m1 + 1, e->run_off);
      p->h2 = e->pa.h2; p->out = e->gen_action; p->ldo = e->Ap;
      fl += fl2 + 2.0 * rows * (double)e->H * A;
    }
    if (pend && !pend_late) {
      if ((rc = pc_tail(&tb.p[np++]))) return rc;
      fl += fl2;
    }
    if (np && (rc = slot(e, "x3_tail", fl, s, [&] { return x3tail_launch(tb, np, s); }))) return rc;
  } else {
  {  // ---- L2
    Group g(e, GEMM_FWD, 0, 0);
    if (value_side) {
      FwdSpec f{TPOL, 2, e->tp.h1, Hp, 0, Hp};
      f.C = e->tp.h2; f.ldc = Hp; f.c_f32 = 0; f.relu = 1; f.mask_idx = -1;
      g.flops += fill_fwd(e, f, rows, g.add());
      for (int c = 0; c < nc; ++c) {
        FwdSpec fc{VAL[c], 2, e->cv[c].h1, Hp, 0, Hp};
        fc.C = e->cv[c].h2; fc.ldc = Hp; fc.c_f32 = 0; fc.relu = 1; fc.mask_idx = 2 * c + 1;
        g.flops += fill_fwd(e, fc, rows, g.add());
      }
    }
    if (actor_side) {
      FwdSpec f{POL, 2, e->pa.h1, Hp, 0, Hp};
      f.C = e->pa.h2; f.ldc = Hp; f.c_f32 = 0; f.relu = 1; f.mask_idx = actor_m1 + 1;
      g.flops += fill_fwd(e, f, rows, g.add());
    }
    int pend_idx = -1;
    if (pend && g.L.nprob < GEMM_MAX_GROUP) {   // the deferred policy-loss critic: -mean Q from the epilogue's partial dots (b3 included)
      const Net& v = e->net[RECNN_NET_VALUE1];
      FwdSpec f{RECNN_NET_VALUE1, 2, e->pc.h1, Hp, 0, Hp};
      f.C = e->pc.h2; f.ldc = Hp; f.c_f32 = 0; f.relu = 1; f.mask_idx = m0 + 1;
      f.dot_w = v.p + v.off[W3]; f.dot_bias = v.p + v.off[B3]; f.dot_part = e->pl_part_base + (int64_t)pp.slot * e->pl_cap;
      pend_idx = g.L.nprob;
      GemmProb* p = g.add();
      g.flops += fill_fwd(e, f, rows, p);
      p->step_add = pp.run_off;
    }
    if ((rc = g.run(s, "fwd_l2"))) return rc;
    if (pend_idx >= 0) {
      const int parts = g.L.batch.p[pend_idx].dot_parts;
      RECNN_REQUIRE(parts > 0 && parts <= e->pl_cap, "policy loss: %d partial sums do not fit %d", parts, e->pl_cap);
      if (pp.slot < LOSS_HIST_MAX) { e->hist_pol_count[pp.slot] = parts; e->hist_pol_add[pp.slot] = 0; }
    }
  }
  {  // ---- L3 of the actors: next_action into the action slot of the packed next rows, gen_action
    Group g(e, GEMM_FWD, 0, 0);
    if (value_side) {
      FwdSpec f{TPOL, 3, e->tp.h2, Hp, 0, Hp};
      f.C = e->xcn; f.ldc = e->ldx; f.c_f32 = 0; f.relu = 0; f.mask_idx = -1;
      if (e->td3) { f.addend = e->ext_noise ? e->ext_noise : e->noise_buf; f.ld_add = A; f.add_clip = e->hy.noise_clip; }
      g.flops += fill_fwd(e, f, rows, g.add());
    }
    if (actor_side) {
      FwdSpec f{POL, 3, e->pa.h2, Hp, 0, Hp};
      f.C = e->gen_action; f.ldc = e->Ap; f.c_f32 = 0; f.relu = 0; f.mask_idx = -1;
      g.flops += fill_fwd(e, f, rows, g.add());
    }
    if ((rc = g.run(s, "fwd_l3_actors"))) return rc;
  }
  }
  e->panel_bwd_done = false;
  e->half_panels = false;
  e->unit_bwd = false;
  if (!value_side) return 0;
  {  // ---- target critics: the action part on top of the state part
    Group g(e, GEMM_FWD, 0, 0);
    for (int c = 0; c < nc; ++c) {
      FwdSpec f{TVAL[c], 1, e->xcn, e->ldx, 0, e->Ap};
      f.C = e->tq[c].h1; f.ldc = Hp; f.c_f32 = 0; f.relu = 1; f.mask_idx = -1;
      f.addend = e->tc_part[c]; f.ld_add = ldp; f.add_clip = INFINITY;
      g.flops += 2.0 * rows * (double)e->H * A;
      fill_fwd(e, f, rows, g.add());
    }
    if (pend_late) pc_l1(g);
    if ((rc = g.run(s, "fwd_l1_target_critic"))) return rc;
  }
  if (tails) {   // Q'(s', a') per row: layer 2 and the last layer's dot in one panel launch
    X3TailBatch tb;
    for (int c = 0; c < nc; ++c) {
      fill_x3tail(e, &tb.p[c], TVAL[c], rows, e->tq[c].h1, -1, e->run_off);
      tb.p[c].q = e->tqv[c];
    }
    int np = nc;
    if (pend_late && (rc = pc_tail(&tb.p[np++]))) return rc;
    if ((rc = slot(e, "x3_tail_target_critic", np * 2.0 * rows * (double)e->H * e->H, s, [&] { return x3tail_launch(tb, np, s); }))) return rc;
  } else {
    Group g(e, GEMM_FWD, 0, 0);
    for (int c = 0; c < nc; ++c) {
      FwdSpec f{TVAL[c], 2, e->tq[c].h1, Hp, 0, Hp};
      f.C = e->tq[c].h2; f.ldc = Hp; f.c_f32 = 0; f.relu = 1; f.mask_idx = -1;
      g.flops += fill_fwd(e, f, rows, g.add());
    }
    if ((rc = g.run(s, "fwd_l2_target_critic"))) return rc;
  }
  {  // heads: TD target, Q, dQ, loss partials (+ dz2 and the last layer's gradient partials)
    HeadArgs h;
    memset(&h, 0, sizeof(h));
    h.rows = rows; h.H = e->H; h.tc_bf16 = e->cfg.dtype; h.ld_h = Hp;
    h.n_target = nc;
    for (int c = 0; c < nc; ++c) {
      const Net& t = e->net[TVAL[c]];
      h.th2[c] = e->tq[c].h2; h.tw3[c] = t.p + t.off[W3]; h.tb3[c] = t.p + t.off[B3];
      if (tails) h.tq_in[c] = e->tqv[c];
      const Net& v = e->net[VAL[c]];
      h.ch2[c] = e->cv[c].h2; h.cw3[c] = v.p + v.off[W3]; h.cb3[c] = v.p + v.off[B3];
      h.q[c] = e->q[c]; h.delta[c] = e->delta[c]; h.loss_part[c] = e->loss_part[c];
    }
    h.reward = e->reward; h.done = e->done;
    h.gamma = e->hy.gamma;
    h.lo = e->td3 ? -INFINITY : e->hy.min_value;
    h.hi = e->td3 ? INFINITY : e->hy.max_value;
    h.expected = e->expected; h.target_q = e->target_q;
    h.n_critic = nc;
    h.policy_mode = 0;
    h.do_bwd = value_bwd;
    h.train = e->cfg.mask_mode != RECNN_MASK_NONE;
    if (value_bwd) {
      for (int c = 0; c < nc; ++c) {
        Net& v = e->net[VAL[c]];
        RECNN_REQUIRE(v.g, "value backward: network %d has no gradient arena bound", VAL[c]);
        h.dz2[c] = e->dzc2[c]; h.dw3_part[c] = v.gp[W3]; h.db2_part[c] = v.gp[B2]; h.db3_part[c] = v.gp[B3];
      }
    }
    {
      const GatherArgs* hg = (value_bwd && e->head_gather) ? e->head_gather : nullptr;   // (armed by step_impl: the next step's gather rides here)
      if ((rc = slot(e, hg ? "head_td_target+gather" : "head_td_target", 0, s, [&] { return head_launch(h, s, hg); }, hg == nullptr))) return rc;
      if (hg) e->head_gather = nullptr;         // consumed
    }
  }
  return 0;
}

// Forward of the value side (+ optionally the actor forward, which is independent of it).
int ph_forward(recnn_engine* e, int rows, bool value_side, bool actor_side, bool value_bwd, hipStream_t s) {
  const int A = e->A, Hp = e->Hp, nc = e->n_critic;
  const int POL = RECNN_NET_POLICY, TPOL = RECNN_NET_TARGET_POLICY;
  const int VAL[2] = {RECNN_NET_VALUE1, RECNN_NET_VALUE2}, TVAL[2] = {RECNN_NET_TARGET_VALUE1, RECNN_NET_TARGET_VALUE2};
  const int actor_m1 = e->td3 ? 4 : 2;  // external mask index of the actor's first dropout
  int rc;
  if (e->x3) return ph_forward_x3(e, rows, value_side, actor_side, value_bwd, s);
  if (e->tune.split_fwd >= 2 && value_chain_ok(e) && e->tune.bwd_panel >= 2 && e->H % 8 == 0) return ph_forward_split(e, rows, value_side, actor_side, value_bwd, s);
  bool chained = false;  // target critics computed inside the first fused launch
  bool fwd_did_bwd = false;  // ... and the critics' head + layer-2 backward too
  // chained target critics add nc producer problems, so a value-side launch always has >= 3 problems
  const bool can_chain = value_side && value_chain_ok(e);
  const int n_first = (value_side ? 1 + nc : 0) + (actor_side ? 1 : 0) + (can_chain ? nc : 0);
  if (fused_mlp_ok(e, n_first)) {
    // whole networks per launch: {target actor, critic(s), actor}
    if (value_side && e->td3 && !e->ext_noise) {
      if ((rc = slot(e, "td3_noise", 0, s, [&] { return noise_fill_launch(e->noise_buf, (int64_t)rows * A, e->hy.noise_std, e->cfg.seed, e->counters, e->run_off, s); }))) return rc;
    }
    const int64_t aoff = tc_off(e, A);
    {
      MlpBatch mb;
      memset(&mb, 0, sizeof(mb));
      int np = 0;
      double fl = 0;
      chained = can_chain;
      const bool in_fwd_bwd = chained && e->tune.bwd_panel >= 2 && mlp_waves() == 16;
      fwd_did_bwd = in_fwd_bwd;
      if (chained) {
        // producers first (launch order = dispatch order): state part of each target critic's layer 1
        for (int c = 0; c < nc; ++c) {
          MlpSpec fp{TVAL[c], e->xcn + aoff, e->ldx, e->K1a, A};
          MlpProb* p = &mb.p[np++];
          fill_mlp(e, fp, rows, p);
          p->W2 = nullptr; p->W3 = nullptr; p->q = nullptr;
          p->part_out = e->tc_part[c]; p->part_flag = e->tc_flag[c];
          fl += 2.0 * rows * (double)e->H * e->S;
        }
      }
      if (value_side) {
        // launch order = dispatch order, and a workgroup may only wait for workgroups dispatched before it: the critics
        // (whose Q(s, a) the head in the target actor's workgroup picks up) come BEFORE the target actor, like the
        // target critics' layer-1 producers -- otherwise a batch with more panels than CUs would dead-lock (bounded)
        for (int c = 0; c < nc; ++c) {
          MlpSpec fc{VAL[c], e->xcs, e->ldx, e->K1c, 0};
          fc.h1 = e->cv[c].h1; fc.h2 = e->cv[c].h2; fc.mask_idx = 2 * c;
          fc.q = e->q[c];
          MlpProb* pc = &mb.p[np];
          fl += fill_mlp(e, fc, rows, &mb.p[np++]);
          if (in_fwd_bwd) {
            MlpCriticBwd& B = mb.cbwd[c];
            pc->cbwd_idx = c;
            B.enabled = 1;
            B.q_slot = e->q_slot[c];
            B.scale = e->cfg.mask_mode != RECNN_MASK_NONE ? 2.0f : 1.0f;
            B.dz2 = e->dzc2[c]; B.dz1 = e->dzc1[c];   // UNIT tensors: the dW launch applies the per-row seed e->delta[c]
            if (value_bwd) RECNN_REQUIRE(e->net[VAL[c]].g, "value backward: network %d has no gradient arena bound", VAL[c]);
            fl += 2.0 * rows * (double)e->H * e->H;
          }
        }
        MlpSpec f{TPOL, e->xcn + aoff, e->ldx, e->K1a, 0};
        f.out = e->xcn; f.ldo = e->ldx;
        if (e->td3) { f.addend = e->ext_noise ? e->ext_noise : e->noise_buf; f.ld_add = A; f.add_clip = e->hy.noise_clip; }
        MlpProb* pt = &mb.p[np];
        fl += fill_mlp(e, f, rows, &mb.p[np++]);
        if (chained && in_fwd_bwd) {
          MlpHead& Hd = mb.head;
          Hd.n_critic = nc;
          for (int c = 0; c < nc; ++c) {
            Hd.q_slot[c] = e->q_slot[c];
            Hd.delta_out[c] = e->delta[c]; Hd.loss_part[c] = e->loss_part[c];
            Hd.db3_part[c] = value_bwd ? e->net[VAL[c]].gp[B3] : nullptr;
          }
          Hd.reward = e->reward; Hd.done = e->done; Hd.gamma = e->hy.gamma;
          Hd.lo = e->td3 ? -INFINITY : e->hy.min_value;
          Hd.hi = e->td3 ? INFINITY : e->hy.max_value;
          Hd.expected = e->expected; Hd.target_q = e->target_q;
        }
        if (chained) {
          pt->n_tail = nc;
          for (int c = 0; c < nc; ++c) {
            const Net& t = e->net[TVAL[c]];
            MlpTail& T = mb.tail[c];
            T.part = e->tc_part[c]; T.flag = e->tc_flag[c];
            T.W1a = sh_ptr(e, TVAL[c], W1); T.ldw1 = t.ld_w1;
            T.W2 = sh_ptr(e, TVAL[c], W2); T.ldw2 = t.ld_w2;
            T.b1 = t.p + t.off[B1]; T.b2 = t.p + t.off[B2]; T.b3 = t.p + t.off[B3]; T.w3row = t.p + t.off[W3];
            T.q = e->tqv[c];
            fl += 2.0 * rows * ((double)e->H * A + (double)e->H * e->H + e->H);
          }
        }
      }
      if (actor_side) {
        MlpSpec f{POL, e->xcs + aoff, e->ldx, e->K1a, 0};
        f.h1 = e->pa.h1; f.h2 = e->pa.h2; f.out = e->gen_action; f.ldo = e->Ap; f.mask_idx = actor_m1;
        fl += fill_mlp(e, f, rows, &mb.p[np++]);
      }
      if (e->pending_pc.on && np < MLP_MAX_GROUP) {
        // the policy-loss forward of the PREVIOUS step of this run (critic on [gen_action | state] of that step's
        // batch, weights as that step's optimizer left them = the current ones) rides here as one more problem: nothing
        // of this or later steps depends on it, and the launch has idle CUs once its short workgroups are done
        const auto& pp = e->pending_pc;
        MlpSpec f{RECNN_NET_VALUE1, pp.ga, e->Ap, e->Ap, 0};
        f.A1 = pp.xs + aoff; f.lda1 = e->ldx; f.K1 = e->K1a; f.col1 = A;
        f.q = e->pl_part_base + (int64_t)pp.slot * e->pl_cap;   // that step's policy-loss slot: Q per row, b3 included
        f.mask_idx = e->td3 ? 6 : 4;
        MlpProb* pd = &mb.p[np++];
        fl += fill_mlp(e, f, rows, pd);
        pd->step_add = pp.run_off;
        e->pending_pc.on = false;
      }
      mb.err = (int32_t*)(e->losses + 4);
      if ((rc = slot(e, "mlp_fwd_nets", fl, s, [&] { return mlp_launch(mb, np, s); }))) return rc;
    }
  } else {
  {  // layer 1: packed rows (compute type) in, tc hidden out
    Group g(e, GEMM_FWD, 0, 0);
    if (value_side) {
      FwdSpec f{TPOL, 1, e->xcn + tc_off(e, A), e->ldx, 0, e->K1a};
      f.C = e->tp.h1; f.ldc = Hp; f.c_f32 = 0; f.relu = 1; f.mask_idx = -1;
      g.flops += fill_fwd(e, f, rows, g.add());
      for (int c = 0; c < nc; ++c) {
        FwdSpec fc{VAL[c], 1, e->xcs, e->ldx, 0, e->K1c};
        fc.C = e->cv[c].h1; fc.ldc = Hp; fc.c_f32 = 0; fc.relu = 1; fc.mask_idx = 2 * c;
        g.flops += fill_fwd(e, fc, rows, g.add());
      }
    }
    if (actor_side) {
      FwdSpec f{POL, 1, e->xcs + tc_off(e, A), e->ldx, 0, e->K1a};
      f.C = e->pa.h1; f.ldc = Hp; f.c_f32 = 0; f.relu = 1; f.mask_idx = actor_m1;
      g.flops += fill_fwd(e, f, rows, g.add());
    }
    if ((rc = g.run(s, "fwd_l1"))) return rc;
  }
  {  // layer 2
    Group g(e, GEMM_FWD, 0, 0);
    if (value_side) {
      FwdSpec f{TPOL, 2, e->tp.h1, Hp, 0, Hp};
      f.C = e->tp.h2; f.ldc = Hp; f.c_f32 = 0; f.relu = 1; f.mask_idx = -1;
      g.flops += fill_fwd(e, f, rows, g.add());
      for (int c = 0; c < nc; ++c) {
        FwdSpec fc{VAL[c], 2, e->cv[c].h1, Hp, 0, Hp};
        fc.C = e->cv[c].h2; fc.ldc = Hp; fc.c_f32 = 0; fc.relu = 1; fc.mask_idx = 2 * c + 1;
        g.flops += fill_fwd(e, fc, rows, g.add());
      }
    }
    if (actor_side) {
      FwdSpec f{POL, 2, e->pa.h1, Hp, 0, Hp};
      f.C = e->pa.h2; f.ldc = Hp; f.c_f32 = 0; f.relu = 1; f.mask_idx = actor_m1 + 1;
      g.flops += fill_fwd(e, f, rows, g.add());
    }
    if ((rc = g.run(s, "fwd_l2"))) return rc;
  }
  if (value_side && e->td3 && !e->ext_noise) {
    if ((rc = slot(e, "td3_noise", 0, s, [&] { return noise_fill_launch(e->noise_buf, (int64_t)rows * A, e->hy.noise_std, e->cfg.seed, e->counters, e->run_off, s); }))) return rc;
  }
  {  // layer 3 of the actors: next_action into the action slot of the packed next rows, gen_action
    Group g(e, GEMM_FWD, 0, 0);
    if (value_side) {
      FwdSpec f{TPOL, 3, e->tp.h2, Hp, 0, Hp};
      f.C = e->xcn; f.ldc = e->ldx; f.c_f32 = 0; f.relu = 0; f.mask_idx = -1;
      if (e->td3) {  // td3.py:74-78: next_action += clamp(noise)
        f.addend = e->ext_noise ? e->ext_noise : e->noise_buf;
        f.ld_add = A;
        f.add_clip = e->hy.noise_clip;
      }
      g.flops += fill_fwd(e, f, rows, g.add());
    }
    if (actor_side) {
      FwdSpec f{POL, 3, e->pa.h2, Hp, 0, Hp};
      f.C = e->gen_action; f.ldc = e->Ap; f.c_f32 = 0; f.relu = 0; f.mask_idx = -1;
      g.flops += fill_fwd(e, f, rows, g.add());
    }
    if ((rc = g.run(s, "fwd_l3_actors"))) return rc;
  }
  }  // first group
  if (chained) {
    // nothing left to launch for the target critics
  } else if (value_side && fused_mlp_ok(e, nc)) {
    {
      MlpBatch mb;
      memset(&mb, 0, sizeof(mb));
      double fl = 0;
      for (int c = 0; c < nc; ++c) {
        MlpSpec f{TVAL[c], e->xcn, e->ldx, e->K1c, 0};
        f.h2 = e->tq[c].h2;
        fl += fill_mlp(e, f, rows, &mb.p[c]);
      }
      if ((rc = slot(e, "mlp_fwd_target_critic", fl, s, [&] { return mlp_launch(mb, nc, s); }))) return rc;
    }
  } else if (value_side) {
    {  // target critics on [next_action | next_state]
      Group g(e, GEMM_FWD, 0, 0);
      for (int c = 0; c < nc; ++c) {
        FwdSpec f{TVAL[c], 1, e->xcn, e->ldx, 0, e->K1c};
        f.C = e->tq[c].h1; f.ldc = Hp; f.c_f32 = 0; f.relu = 1; f.mask_idx = -1;
        g.flops += fill_fwd(e, f, rows, g.add());
      }
      if ((rc = g.run(s, "fwd_l1_target_critic"))) return rc;
    }
    {
      Group g(e, GEMM_FWD, 0, 0);
      for (int c = 0; c < nc; ++c) {
        FwdSpec f{TVAL[c], 2, e->tq[c].h1, Hp, 0, Hp};
        f.C = e->tq[c].h2; f.ldc = Hp; f.c_f32 = 0; f.relu = 1; f.mask_idx = -1;
        g.flops += fill_fwd(e, f, rows, g.add());
      }
      if ((rc = g.run(s, "fwd_l2_target_critic"))) return rc;
    }
  }
  e->panel_bwd_done = false;
  e->half_panels = false;
  e->unit_bwd = false;
  if (value_side && fwd_did_bwd) {
    // nothing left to launch here: losses, the per-row seed and the UNIT dz2 / dz1 came out of the forward launch; the dW
    // launch scales them and adds the bias / last-layer partial sums
    e->panel_bwd_done = true;
    e->unit_bwd = true;
  } else if (value_side && chained && e->tune.bwd_panel) {
    // critic head + dz2 + dz1 in one row-panel launch (bwd.hip); Q comes from the forward launch, Q' from its tails
    BwdPanelBatch bb;
    memset(&bb, 0, sizeof(bb));
    const bool train = e->cfg.mask_mode != RECNN_MASK_NONE;
    double fl = 0;
    for (int c = 0; c < nc; ++c) {
      BwdPanelProb& b = bb.p[c];
      Net& v = e->net[VAL[c]];
      b.rows = rows; b.H = e->H; b.mode = 0;
      b.q = e->q[c]; b.n_target = nc;
      for (int t = 0; t < nc; ++t) b.tq[t] = e->tqv[t];
      b.reward = e->reward; b.done = e->done; b.gamma = e->hy.gamma;
      b.lo = e->td3 ? -INFINITY : e->hy.min_value;
      b.hi = e->td3 ? INFINITY : e->hy.max_value;
      if (c == 0) { b.expected = e->expected; b.target_q = e->target_q; }
      b.delta_out = e->delta[c]; b.loss_part = e->loss_part[c];
      b.h2 = e->cv[c].h2; b.ldh = Hp; b.w3 = v.p + v.off[W3]; b.scale = train ? 2.0f : 1.0f;
      b.dz2 = e->dzc2[c];
      b.W2 = sh_ptr(e, VAL[c], W2); b.ldw2 = v.ld_w2;
      b.h1 = e->cv[c].h1; b.dz1 = e->dzc1[c];
      if (value_bwd) {
        RECNN_REQUIRE(v.g, "value backward: network %d has no gradient arena bound", VAL[c]);
        b.dw3_part = v.gp[W3]; b.db2_part = v.gp[B2]; b.db3_part = v.gp[B3]; b.colsum = v.gp[B1];
      }
      fl += 2.0 * rows * (double)e->H * e->H;
    }
    if ((rc = slot(e, "head_dx_critic", fl, s, [&] { return bwd_panel_launch(bb, nc, s); }))) return rc;
    e->panel_bwd_done = true;
  } else if (value_side) {
    // heads: TD target, Q, dQ, loss partials
    HeadArgs h;
    memset(&h, 0, sizeof(h));
    h.rows = rows; h.H = e->H; h.tc_bf16 = e->cfg.dtype; h.ld_h = Hp;
    h.n_target = nc;
    for (int c = 0; c < nc; ++c) {
      const Net& t = e->net[TVAL[c]];
      h.th2[c] = e->tq[c].h2; h.tw3[c] = t.p + t.off[W3]; h.tb3[c] = t.p + t.off[B3];
      if (chained) h.tq_in[c] = e->tqv[c];
      const Net& v = e->net[VAL[c]];
      h.ch2[c] = e->cv[c].h2; h.cw3[c] = v.p + v.off[W3]; h.cb3[c] = v.p + v.off[B3];
      h.q[c] = e->q[c]; h.delta[c] = e->delta[c]; h.loss_part[c] = e->loss_part[c];
    }
    h.reward = e->reward; h.done = e->done;
    h.gamma = e->hy.gamma;
    h.lo = e->td3 ? -INFINITY : e->hy.min_value;
    h.hi = e->td3 ? INFINITY : e->hy.max_value;
    h.expected = e->expected; h.target_q = e->target_q;
    h.n_critic = nc;
    h.policy_mode = 0;
    h.do_bwd = value_bwd;
    h.train = e->cfg.mask_mode != RECNN_MASK_NONE;
    if (value_bwd) {
      for (int c = 0; c < nc; ++c) {
        Net& v = e->net[VAL[c]];
        RECNN_REQUIRE(v.g, "value backward: network %d has no gradient arena bound", VAL[c]);
        h.dz2[c] = e->dzc2[c]; h.dw3_part[c] = v.gp[W3]; h.db2_part[c] = v.gp[B2]; h.db3_part[c] = v.gp[B3];
      }
    }
    {
      const GatherArgs* hg = (value_bwd && e->head_gather) ? e->head_gather : nullptr;   // (armed by step_impl: the next step's gather rides here)
      if ((rc = slot(e, hg ? "head_td_target+gather" : "head_td_target", 0, s, [&] { return head_launch(h, s, hg); }, hg == nullptr))) return rc;
      if (hg) e->head_gather = nullptr;         // consumed
    }
  }
  return 0;
}

// Backward of the critic(s): dz1 (unless the forward launch produced it), then the weight-gradient GEMMs as split-batch slabs
// (summed by the slab-reducing Adam launch, or by grad_reduce into the flat arenas when `reduce`: the phase API / data parallel).
int ph_value_backward(recnn_engine* e, int rows, bool reduce, hipStream_t s, bool dx_only) {
  const int Hp = e->Hp, H = e->H, nc = e->n_critic;
  const int VAL[2] = {RECNN_NET_VALUE1, RECNN_NET_VALUE2};
  int rc;
  if (!e->panel_bwd_done) {
    Group g(e, GEMM_DX, 0, 0);
    for (int c = 0; c < nc; ++c)
      g.flops += fill_dx(e, g.add(), rows, e->dzc2[c], Hp, Hp, VAL[c], W2, 0, H, e->dzc1[c], Hp, e->cv[c].h1, Hp, e->net[VAL[c]].gp[B1]);
    if ((rc = g.run(s, "dx_critic_l2"))) return rc;
  }
  if (dx_only) return 0;     // (the weight gradients follow in dw_adam_kernel: ph_value_dwadam)
  NetLayout L0 = make_layout(e, VAL[0], rows);
  {
    Group g(e, GEMM_DW, 0, 0);  // dW2 = dz2^T h1 and dW1 = dz1^T [a|s], split over the batch into slabs
    DwVec vec;
    memset(&vec, 0, sizeof(vec));
    for (int c = 0; c < nc; ++c) {
      GemmProb* p = g.add();
      g.flops += fill_dw(e, p, rows, e->dzc2[c], Hp, H, e->cv[c].h1, Hp, H, 0, e->net[VAL[c]].gp[W2], L0.t[W2].nslab,
                         L0.t[W2].slab_stride);
      if (e->unit_bwd) p->a_row_scale = e->delta[c];
    }
    for (int c = 0; c < nc; ++c) {
      GemmProb* p = g.add();
      g.flops += fill_dw(e, p, rows, e->dzc1[c], Hp, H, e->xcs, e->ldx, e->S + e->A, e->S, e->net[VAL[c]].gp[W1],
                         L0.t[W1].nslab, L0.t[W1].slab_stride);
      if (e->unit_bwd) p->a_row_scale = e->delta[c];
    }
    if (e->unit_bwd) {  // dW3 / db2 / db1 partial sums per 32-row panel ride on this launch
      vec.n = nc;
      for (int c = 0; c < nc; ++c) {
        Net& v = e->net[VAL[c]];
        DwVecProb& q = vec.p[c];
        q.rows = rows; q.H = H; q.delta = e->delta[c]; q.h2 = e->cv[c].h2; q.u2 = e->dzc2[c]; q.U = e->dzc1[c]; q.ldh = Hp;
        q.dw3_part = v.gp[W3]; q.db2_part = v.gp[B2]; q.colsum = v.gp[B1];
      }
      g.L.vec = &vec;
    }
    if ((rc = g.run(s, "dw_critic"))) return rc;
  }
  for (int c = 0; c < nc && reduce; ++c) {
    NetLayout L = make_layout(e, VAL[c], rows);
    if ((rc = slot(e, "grad_reduce_critic", 0, s, [&] { return grad_reduce_launch(L, g_produce(e, VAL[c]), nullptr, s); }))) return rc;
  }
  return 0;
}

// Policy loss through the (updated) critic 1; optionally the gradient chain back into the actor.
// need_rows: per-row Q(s, pi(s)) wanted (debug / learn=False logging) -> head kernel; otherwise, on the fused bf16 path,
// the loss is summed in the layer-2 GEMM's epilogue and the backward seed comes from the row-panel kernel (bwd.hip).
int ph_policy(recnn_engine* e, int rows, bool backward, bool with_l1, hipStream_t s, bool need_rows) {
  const int A = e->A, Hp = e->Hp, H = e->H, Ap = e->Ap;
  const int POL = RECNN_NET_POLICY, V1 = RECNN_NET_VALUE1;
  const int m0 = e->td3 ? 6 : 4;
  const bool train = e->cfg.mask_mode != RECNN_MASK_NONE;
  int rc;
  // (split bf16: the loss of a step without backward comes from the layer-2 epilogue's partial dots; with backward the head kernel
  // produces the seed dz_e2 as well -- either way the step's partial sums land in its pl_part slot)
  const bool x3_dot = e->x3 && !need_rows && !backward;
  const bool use_dot = (!need_rows && value_panel_ok(e) && !fused_mlp_ok(e, 1)) || x3_dot;
  bool chain_done = false;
  e->pl_dot_parts = 0;
  if (fused_mlp_ok(e, 1)) {
    // critic on [gen_action | state] with the UPDATED weights: one launch, two layer-1 contraction segments
    MlpBatch mb;
    memset(&mb, 0, sizeof(mb));
    MlpSpec f{V1, e->gen_action, Ap, Ap, 0};
    f.A1 = e->xcs + tc_off(e, A); f.lda1 = e->ldx; f.K1 = e->K1a; f.col1 = A;
    f.h1 = e->pc.h1; f.h2 = e->pc.h2; f.mask_idx = m0;
    const double fl = fill_mlp(e, f, rows, &mb.p[0]);
    if ((rc = slot(e, "mlp_fwd_pcritic", fl, s, [&] { return mlp_launch(mb, 1, s); }))) return rc;
  } else {
  {  // critic L1 on [gen_action | state]: two contraction segments over the rotated W1 shadow
    Group g(e, GEMM_FWD, 0, 0);
    FwdSpec f{V1, 1, e->gen_action, Ap, 0, Ap};
    f.b_col = 0;
    f.A2 = e->xcs + tc_off(e, A); f.lda2 = e->ldx; f.K2 = e->K1a; f.b2_col = A;
    f.C = e->pc.h1; f.ldc = Hp; f.c_f32 = 0; f.relu = 1; f.mask_idx = m0;
    // gen_action is zero padded to Ap columns, so segment 0 may run over the padded width: the W1
    // shadow columns it meets there (the first state columns) are multiplied by zeros.
    g.flops += fill_fwd(e, f, rows, g.add());
    if ((rc = g.run(s, "fwd_l1_pcritic"))) return rc;
  }
  {
    Group g(e, GEMM_FWD, 0, 0);
    FwdSpec f{V1, 2, e->pc.h1, Hp, 0, Hp};
    f.C = e->pc.h2; f.ldc = Hp; f.c_f32 = 0; f.relu = 1; f.mask_idx = m0 + 1;
    if (use_dot) { f.dot_w = e->net[V1].p + e->net[V1].off[W3]; f.dot_bias = e->net[V1].p + e->net[V1].off[B3]; f.dot_part = e->pl_part; }
    g.flops += fill_fwd(e, f, rows, g.add());
    if ((rc = g.run(s, "fwd_l2_pcritic"))) return rc;
    if (use_dot) {
      e->pl_dot_parts = g.L.batch.p[0].dot_parts;
      RECNN_REQUIRE(e->pl_dot_parts > 0 && e->pl_dot_parts <= e->pl_cap, "policy loss: %d partial sums do not fit %d", e->pl_dot_parts, e->pl_cap);
      if (e->run_off < LOSS_HIST_MAX) { e->hist_pol_count[e->run_off] = e->pl_dot_parts; e->hist_pol_add[e->run_off] = 0; }
    }
  }
  }
  if (use_dot) {
    if (!backward) return 0;
    Net& pn0 = e->net[POL];
    RECNN_REQUIRE(pn0.g, "policy backward: the actor has no gradient arena bound");
    if (e->tune.policy_chain) {
      // the whole chain dz_e2 -> dz_e1 -> dact -> dz_p2 -> dz_p1 on a row panel that never leaves the CU (bwd.hip)
      BwdChainArgs c;
      memset(&c, 0, sizeof(c));
      const Net& v = e->net[V1];
      c.rows = rows; c.H = H; c.A = A; c.delta_const = -1.0f / (float)rows; c.scale = train ? 2.0f : 1.0f; c.ldh = Hp;
      c.e2 = e->pc.h2; c.e1 = e->pc.h1; c.w3c = v.p + v.off[W3];
      c.W2c = sh_ptr(e, V1, W2); c.ldw2c = v.ld_w2;
      c.W1c = sh_ptr(e, V1, W1); c.ldw1c = v.ld_w1;
      c.W3a = sh_ptr(e, POL, W3); c.ldw3a = pn0.ld_w3;
      c.W2a = sh_ptr(e, POL, W2); c.ldw2a = pn0.ld_w2;
      c.p2 = e->pa.h2; c.p1 = e->pa.h1;
      c.dact = e->dag; c.ldact = Ap; c.dzp2 = e->dzp2; c.dzp1 = e->dzp1;
      c.db3_part = pn0.gp[B3]; c.db2_part = pn0.gp[B2]; c.db1_part = pn0.gp[B1];
      const double fl = 2.0 * rows * ((double)H * H + (double)H * A + (double)A * H + (double)H * H);
      if ((rc = slot(e, "bwd_chain_policy", fl, s, [&] { return bwd_chain_launch(c, s); }))) return rc;
      chain_done = true;
    } else {
    // backward seed d = -1/B for every row: dz_e2 and dz_e1 in one row-panel launch, no critic parameter gradients
    BwdPanelBatch bb;
    memset(&bb, 0, sizeof(bb));
    BwdPanelProb& b = bb.p[0];
    const Net& v = e->net[V1];
    b.rows = rows; b.H = H; b.mode = 1; b.delta_const = -1.0f / (float)rows;
    b.h2 = e->pc.h2; b.ldh = Hp; b.w3 = v.p + v.off[W3]; b.scale = train ? 2.0f : 1.0f;
    b.dz2 = e->dze2; b.W2 = sh_ptr(e, V1, W2); b.ldw2 = v.ld_w2; b.h1 = e->pc.h1; b.dz1 = e->dze1;
    if ((rc = slot(e, "head_dx_pcritic", 2.0 * rows * (double)H * H, s, [&] { return bwd_panel_launch(bb, 1, s); }))) return rc;
    }
  } else {
    HeadArgs h;
    memset(&h, 0, sizeof(h));
    const Net& v = e->net[V1];
    h.rows = rows; h.H = H; h.tc_bf16 = e->cfg.dtype; h.ld_h = Hp;
    h.n_target = 0; h.n_critic = 1; h.policy_mode = 1;
    h.ch2[0] = e->pc.h2; h.cw3[0] = v.p + v.off[W3]; h.cb3[0] = v.p + v.off[B3];
    h.q[0] = e->qpi; h.loss_part[0] = e->loss_part[2];
    if (e->x3 && !need_rows) {    // the step's policy-loss partial sums (sum of Q incl. b3 per block) in its pl_part slot
      const int nblk = (rows + HEAD_ROWS_PER_BLOCK - 1) / HEAD_ROWS_PER_BLOCK;
      h.loss_part[0] = e->pl_part;
      e->pl_dot_parts = nblk;
      if (e->run_off < LOSS_HIST_MAX) { e->hist_pol_count[e->run_off] = nblk; e->hist_pol_add[e->run_off] = 0; }
    }
    h.do_bwd = backward;          // d(policy_loss)/dQ = -1/B for every row; no critic parameter gradients
    h.train = train;
    h.delta_const = -1.0f / (float)rows;
    h.dz2[0] = e->dze2;
    if ((rc = slot(e, "head_policy_loss", 0, s, [&] { return head_launch(h, s); }))) return rc;
  }
  if (!backward) return 0;
  Net& pn = e->net[POL];
  RECNN_REQUIRE(pn.g, "policy backward: the actor has no gradient arena bound");
  auto dx1 = [&](const char* nm, const void* Ain, int64_t lda, int Kc, int ni, int which, int N, void* C, int64_t ldc, const void* yref,
                 float* colsum) {
    Group g(e, GEMM_DX, 0, 0);
    g.flops += fill_dx(e, g.add(), rows, Ain, lda, Kc, ni, which, 0, N, C, ldc, yref, Hp, colsum);
    return g.run(s, nm);
  };
  // critic: dz_e1 = (dz_e2 W2) * 2[e1>0];   dact = dz_e1 * W1[:, action columns]  (shadow columns 0..A-1)
  if (!chain_done) {
  if (!use_dot && (rc = dx1("dx_pcritic_l2", e->dze2, Hp, Hp, V1, W2, H, e->dze1, Hp, e->pc.h1, nullptr))) return rc;
  if ((rc = dx1("dx_pcritic_action", e->dze1, Hp, Hp, V1, W1, A, e->dag, Ap, nullptr, pn.gp[B3]))) return rc;
  // actor: dz_p2 = (dact W3) * 2[p2>0];  dz_p1 = (dz_p2 W2) * 2[p1>0]
  if ((rc = dx1("dx_actor_l3", e->dag, Ap, Ap, POL, W3, H, e->dzp2, Hp, e->pa.h2, pn.gp[B2]))) return rc;
  if ((rc = dx1("dx_actor_l2", e->dzp2, Hp, Hp, POL, W2, H, e->dzp1, Hp, e->pa.h1, pn.gp[B1]))) return rc;
  }
  NetLayout L = make_layout(e, POL, rows);
  {
    Group g(e, GEMM_DW, 0, 0);
    g.flops += fill_dw(e, g.add(), rows, e->dag, Ap, A, e->pa.h2, Hp, H, 0, pn.gp[W3], L.t[W3].nslab, L.t[W3].slab_stride);
    g.flops += fill_dw(e, g.add(), rows, e->dzp2, Hp, H, e->pa.h1, Hp, H, 0, pn.gp[W2], L.t[W2].nslab, L.t[W2].slab_stride);
    g.flops += fill_dw(e, g.add(), rows, e->dzp1, Hp, H, e->xcs + tc_off(e, A), e->ldx, e->S, 0, pn.gp[W1], L.t[W1].nslab,
                       L.t[W1].slab_stride);
    if ((rc = g.run(s, "dw_actor"))) return rc;
  }
  return slot(e, "grad_reduce_actor", 0, s, [&] { return grad_reduce_launch(L, g_produce(e, RECNN_NET_POLICY), with_l1 ? pn.l1part : nullptr, s); });
}


int ph_finish(recnn_engine* e, int rows, bool ticked_value, bool ticked_policy, hipStream_t s) {
  if (e->run_off >= 0 && e->run_off < LOSS_HIST_MAX) e->hist_half[e->run_off] = e->half_panels;
  if (e->run_skip_finish) return 0;  // inside a run graph: the last step's finalize ticks the counters for the whole run
  LossFinalizeArgs a;
  memset(&a, 0, sizeof(a));
  const int nblk = (rows + HEAD_ROWS_PER_BLOCK - 1) / HEAD_ROWS_PER_BLOCK;
  const int nc = e->n_critic;
  const int nval = (value_panel_ok(e) || e->half_panels) ? (rows + BWD_ROWS - 1) / BWD_ROWS : nblk;
  for (int c = 0; c < nc; ++c) { a.part[c] = e->loss_part[c]; a.n_part[c] = nval; a.scale[c] = 1.0f / (float)rows; a.pair[c] = e->half_panels; }
  a.part[nc] = e->loss_part[2]; a.n_part[nc] = nblk; a.scale[nc] = -1.0f / (float)rows;
  if (e->pl_dot_parts > 0) {  // policy loss = -(sum of the layer-2 epilogue's partial dots, b3 included) / B
    a.part[nc] = e->pl_part; a.n_part[nc] = e->pl_dot_parts;
  }
  a.n = nc + 1;
  a.out = e->losses;
  a.host_out = e->h_stage;
  // run_tick = {1, 1, 1} outside run graphs; the last step of a run adds the whole run's counts
  a.tick_inc[a.n_tick] = e->run_tick[0]; a.tick[a.n_tick++] = e->counters;  // mask-key step
  if (ticked_value) {
    a.tick_inc[a.n_tick] = e->run_tick[1]; a.tick[a.n_tick++] = e->net[RECNN_NET_VALUE1].t_ptr;
    if (e->td3) { a.tick_inc[a.n_tick] = e->run_tick[1]; a.tick[a.n_tick++] = e->net[RECNN_NET_VALUE2].t_ptr; }
  }
  const bool run_final = e->run_tick[0] > 1;  // last step of a run graph: the actor took run_tick[2] steps during the run
  if (run_final ? e->run_tick[2] > 0 : ticked_policy) { a.tick_inc[a.n_tick] = e->run_tick[2]; a.tick[a.n_tick++] = e->net[RECNN_NET_POLICY].t_ptr; }
  if (e->has_sampler && e->use_sampler) { a.wrap_ptr = e->smp.cursor; a.wrap_mod = e->smp.n_batches; a.wrap_inc = e->run_tick[0]; }
  a.ring = e->loss_ring; a.ring_mask = LOSS_RING - 1;
  if (run_final && e->run_off >= 1 && e->run_off < LOSS_HIST_MAX) {
    // losses of the run's earlier steps from their kept partial sums (the step counter is not ticked yet)
    LossHistoryArgs h;
    memset(&h, 0, sizeof(h));
    h.n_steps = e->run_off; h.n = nc + 1;
    for (int i = 0; i < e->run_off; ++i) h.pair_steps |= (unsigned long long)(e->hist_half[i] ? 1 : 0) << i;
    for (int c = 0; c < nc; ++c) { h.part[c] = e->loss_part_base[c]; h.stride[c] = e->loss_part_stride; h.n_part[c] = nval; h.scale[c] = 1.0f / (float)rows; }
    h.scale[nc] = -1.0f / (float)rows;
    if (value_panel_ok(e) || e->x3) {   // policy loss from the layer-2 epilogue's partial dots / the deferred forward's per-row Q
      h.part[nc] = e->pl_part_base; h.stride[nc] = e->pl_cap;
      for (int i = 0; i < e->run_off; ++i) { h.pol_count[i] = e->hist_pol_count[i]; h.pol_add[i] = e->hist_pol_add[i]; }
    } else {                   // ... from the head kernel's per-block partial sums (fp32 / generic path)
      h.part[nc] = e->loss_part_base[2]; h.stride[nc] = e->loss_part_stride;
      for (int i = 0; i < e->run_off; ++i) { h.pol_count[i] = nblk; h.pol_add[i] = 0; }
    }
    h.b3 = e->net[RECNN_NET_VALUE1].p + e->net[RECNN_NET_VALUE1].off[B3];
    h.step_ctr = e->counters; h.ring = e->loss_ring; h.ring_mask = LOSS_RING - 1;
    int hrc = slot(e, "loss_history", 0, s, [&] { return loss_history_launch(h, s); }, false);
    if (hrc) return hrc;
  }
  return slot(e, "loss_finalize", 0, s, [&] { return loss_finalize_launch(a, s); }, false);
}

}  // namespace recnn_eng

namespace recnn_eng {
int ph_policy_l1(recnn_engine* e, hipStream_t s) {
  Net& pn = e->net[RECNN_NET_POLICY];
  NetLayout L = make_layout(e, RECNN_NET_POLICY, 0);
  return slot(e, "l1_norm_actor", 0, s, [&] {
    return l1_blocks_launch(L, g_consume(e, RECNN_NET_POLICY), pn.l1part, s);
  });
}

// rows > 0: fused single-GPU path, Adam sums the gradient slabs itself (no separate reduction launch)
int value_apply(recnn_engine* e, bool soft, float grad_scale, hipStream_t s, int rows) {
  int rc;
  const int VAL[2] = {RECNN_NET_VALUE1, RECNN_NET_VALUE2}, TVAL[2] = {RECNN_NET_TARGET_VALUE1, RECNN_NET_TARGET_VALUE2};
  for (int c = 0; c < e->n_critic; ++c)
    if ((rc = apply_net(e, VAL[c], rows, true, 1, grad_scale, false, soft ? TVAL[c] : -1, e->hy.soft_tau, s, rows > 0)))
      return rc;
  return 0;
}

// The critics' weight gradients with the optimizer step in the epilogue (dwadam.hip): ONE launch instead of ph_value_backward +
// value_apply, no gradient slabs.  Taken when the step's backward tensors came from the split forward's tail (mlpt.hip: dz2 / dz1
// already carry the per-row loss seed, the small tensors' panel sums exist) on the single-GPU bf16 step; bit-identical to the two launches.
bool dwadam_ok(const recnn_engine* e, int rows) {
  if (!e->tune.dw_fuse || e->comm || e->unit_bwd || e->H != 256) return false;
  if (e->x3) {
    // split bf16: the head launch wrote dz2 (times the loss seed) and the small tensors' partial sums; dz1 comes from the dX launch that
    // ph_value_dwadam issues first.  (The generic split-bf16 forward only: Hp = 512 physical.)
    if (e->cfg.dtype != RECNN_BF16X3 || e->panel_bwd_done || e->Hp != 512) return false;
  } else if (e->cfg.dtype != RECNN_BF16 || !e->panel_bwd_done || e->Hp != 256) return false;
  const int VAL[2] = {RECNN_NET_VALUE1, RECNN_NET_VALUE2};
  for (int c = 0; c < e->n_critic; ++c) {
    const Net& n = e->net[VAL[c]];
    if (!n.g || !n.m || !n.v || !n.critic) return false;
    const NetLayout L = make_layout(e, VAL[c], rows);
    if (!dwadam_tensor_ok(L, W1, rows, e->x3) || !dwadam_tensor_ok(L, W2, rows, e->x3)) return false;
    if (!L.t[B1].small || !L.t[B2].small || !L.t[W3].small || !L.t[B3].small) return false;
  }
  return true;
}

int ph_value_dwadam(recnn_engine* e, int rows, bool soft, hipStream_t s) {
  const int VAL[2] = {RECNN_NET_VALUE1, RECNN_NET_VALUE2}, TVAL[2] = {RECNN_NET_TARGET_VALUE1, RECNN_NET_TARGET_VALUE2};
  DwAdamBatch b;
  memset(&b, 0, sizeof(b));
  double fl = 0;
  int rc;
  if (!e->panel_bwd_done && (rc = ph_value_backward(e, rows, false, s, true))) return rc;   // (split bf16: the critics' dX launch)
  for (int c = 0; c < e->n_critic; ++c) {
    DwAdamNet& n = b.n[c];
    n.L = make_layout(e, VAL[c], rows);
    if ((rc = fill_apply_args(e, VAL[c], n.L, true, 1, 1.0f, false, soft ? TVAL[c] : -1, e->hy.soft_tau, &n.a))) return rc;
    n.a.from_slabs = 1;
    n.rows = rows;
    n.w[0].dz = e->dzc2[c]; n.w[0].ldz = e->Hp; n.w[0].x = e->cv[c].h1; n.w[0].ldx = e->Hp; n.w[0].tensor = W2;
    n.w[1].dz = e->dzc1[c]; n.w[1].ldz = e->Hp; n.w[1].x = e->xcs; n.w[1].ldx = e->ldx; n.w[1].tensor = W1;
    fl += 2.0 * rows * (double)e->H * (e->H + e->S + e->A);
  }
  return slot(e, "dwadam_critic", fl, s, [&] { return dwadam_launch(b, e->n_critic, s); }, false);
}

int policy_apply(recnn_engine* e, bool soft, float grad_scale, hipStream_t s, bool have_l1) {
  int rc;
  if (!have_l1 && (rc = ph_policy_l1(e, s))) return rc;
  // TD3 never soft-updates the target policy (td3.py:136-141); DDPG does (ddpg.py:98-100).
  const int tgt = (soft && !e->td3) ? RECNN_NET_TARGET_POLICY : -1;
  return apply_net(e, RECNN_NET_POLICY, 0, true, 0, grad_scale, true, tgt, e->hy.soft_tau, s);
}

// per-step slot of the loss partial sums (run graphs: step i of the run; everything else: slot 0)
void use_hist_slot(recnn_engine* e, int i) {
  for (int c = 0; c < 3; ++c) e->loss_part[c] = e->loss_part_base[c] + (int64_t)i * e->loss_part_stride;
  e->pl_part = e->pl_part_base + (int64_t)i * e->pl_cap;
}

// batch buffer set k (0: the bound / first set, 1: the look-ahead set of bf16 sampler mode)
void use_set(recnn_engine* e, int k) {
  e->cur_set = k;
  if (k == 0) {
    e->xcs = e->twins ? e->xsh : (char*)e->xs;
    e->xcn = e->twins ? e->xnh : (char*)e->xn;
    e->reward = e->reward0; e->done = e->done0;
    e->gen_action = e->gen_action0;
  } else {
    e->xcs = e->xsh2; e->xcn = e->xnh2; e->reward = e->reward2; e->done = e->done2;
    e->gen_action = e->gen_action2;
  }
}
bool lookahead_ok(const recnn_engine* e) {
  return e->has_sampler && e->twins && !e->tune.sampler_f32_rows && (e->smp.users_per_batch <= 1024 || e->smp.plan) && e->tune.pregather;
}

// gather of the batch `cursor_add` steps ahead of the device cursor into buffer set `set`
GatherArgs gather_args(const recnn_engine* e, int rows, int set, int cursor_add) {
  const recnn_sampler& m = e->smp;
  const bool inl = m.users_per_batch <= 1024;   // the gather plans its rows itself: one launch less
  GatherArgs g;
  memset(&g, 0, sizeof(g));
  g.items = m.items; g.ratings = m.ratings; g.user_off = m.user_off; g.users = m.perm;
  g.row_off = inl ? nullptr : m.row_off;
  g.n_users = m.users_per_batch; g.rows = rows; g.frame = m.frame; g.emb = m.emb_dim; g.table = m.table;
  g.state = e->xs + e->A; g.ld_state = e->ldx32;
  g.next_state = e->xn + e->A; g.ld_next = e->ldx32;
  g.action = e->xs; g.ld_action = e->ldx32;
  g.reward = set ? e->reward2 : e->reward0; g.done = set ? e->done2 : e->done0;
  g.cursor = m.cursor; g.cursor_stride = m.users_per_batch;
  g.cursor_add = cursor_add; g.cursor_mod = m.n_batches;
  g.inline_plan = inl;
  if (m.plan && rows <= m.plan_rows) { g.plan = m.plan; g.plan_stride = m.plan_rows; }
  if (e->twins) {  // the compute-type twins of the packed rows are written by the same kernel
    char* hs = set ? e->xsh2 : e->xsh;
    char* hn = set ? e->xnh2 : e->xnh;
    const int acol = e->x3 ? 2 * e->A : e->A;   // (split rows: the state columns start at physical column 2 A)
    g.state_h = (bf16_t*)hs + acol; g.next_h = (bf16_t*)hn + acol; g.action_h = (bf16_t*)hs;
    g.ld_h = e->ldx;
    g.x3 = e->x3;
    // Nothing reads the fp32 rows when the engine samples its own batches in bf16: materialise the batch in
    // the compute type only (recnn_tune_sampler_f32_rows(1) restores the fp32 copies, e.g. for inspection).
    if (!e->tune.sampler_f32_rows) { g.state = nullptr; g.next_state = nullptr; g.action = nullptr; }
  }
  return g;
}

int frame_gather_packed(recnn_engine* e, int rows, hipStream_t s) {
  const recnn_sampler& m = e->smp;
  const bool inl = m.users_per_batch <= 1024 || (m.plan && rows <= m.plan_rows);
  if (!inl) {
    int rc = slot(e, "frame_plan", 0, s, [&] {
      return recnn_frame_plan(m.user_off, m.perm, m.users_per_batch, m.frame, m.row_off, m.cursor, m.users_per_batch, s);
    });
    if (rc) return rc;
  }
  return slot(e, "frame_gather", 0, s, [&] { return frame_gather_launch(gather_args(e, rows, e->cur_set, e->run_off), s); });
}

// Make the step's batch available in the compute type: built by the sampler, or converted from the bound fp32 rows.
int stage_batch(recnn_engine* e, int rows, hipStream_t s) {
  if (e->has_sampler && e->use_sampler) return frame_gather_packed(e, rows, s);
  if (e->bf16)
    return slot(e, "rows_to_bf16", 0, s, [&] {
      return rows_to_bf16_launch(e->xs, e->xn, (bf16_t*)e->xsh, (bf16_t*)e->xnh, rows, e->ldx, s);
    });
  if (e->x3)   // (padding columns of the split rows rest at zero: the workspace is zero-initialised and nothing writes them)
    return slot(e, "rows_to_x3", 0, s, [&] {
      return rows_to_x3_launch(e->xs, e->xn, (bf16_t*)e->xsh, (bf16_t*)e->xnh, rows, e->S + e->A, e->ldx32, e->ldx, s);
    });
  return 0;
}

// ---- cycle mode: batch j of the cycle = rows j * rows .. (j + 1) * rows - 1 of the m_* arrays
void use_mset(recnn_engine* e, int j, int rows) {
  const int64_t r0 = (int64_t)j * rows;
  e->xcs = e->m_xs + r0 * e->ldx * 2; e->xcn = e->m_xn + r0 * e->ldx * 2;
  e->reward = e->m_reward + r0; e->done = e->m_done + r0;
  e->gen_action = e->m_ga + r0 * e->Ap * 2;
  e->pa.h1 = e->m_pa_h1 + r0 * e->Hp * 2; e->pa.h2 = e->m_pa_h2 + r0 * e->Hp * 2;
  for (int c = 0; c < e->n_critic; ++c) e->tqv[c] = e->m_tq[c] + r0;
}
void leave_mset(recnn_engine* e) {
  e->pa = e->pa0;
  for (int c = 0; c < 2; ++c) e->tqv[c] = e->tqv0[c];
  use_set(e, 0);
}
bool cycle_ok(const recnn_engine* e, int rows) {
  return e->tune.split_fwd && lookahead_ok(e) && value_chain_ok(e) && e->tune.bwd_panel >= 2 && e->H % 8 == 0 && rows % 32 == 0 && e->m_xs != nullptr &&
         !e->ext_noise && e->cfg.mask_mode != RECNN_MASK_EXTERNAL;   // (external masks / noise describe ONE batch)
}

// The FROZEN networks on all n * rows rows of the cycle's batches (gathered by ph_gather_cycle into the current copy):
// target actor -> next_action (+ TD3 noise) -> target critics -> Q', and the actor -> gen_action (+ its activations for the
// policy step's backward).  recnn/nn/update/misc.py:28-31, td3.py:73-81 (target side), ddpg.py:66-69 / td3.py:104-110 (actor).
void select_mbuf(recnn_engine* e, int b) {
  e->m_xs = e->m_xs_b[b]; e->m_xn = e->m_xn_b[b]; e->m_reward = e->m_reward_b[b]; e->m_done = e->m_done_b[b];
}
// the batches of run steps run_off0 .. run_off0 + n - 1 into copy `b` of the cycle arrays: one launch
int ph_gather_cycle(recnn_engine* e, int rows, int n, int run_off0, int b, hipStream_t s) {
  GatherArgs g = gather_args(e, rows, 0, run_off0);
  g.state_h = (bf16_t*)e->m_xs_b[b] + e->A; g.next_h = (bf16_t*)e->m_xn_b[b] + e->A; g.action_h = (bf16_t*)e->m_xs_b[b];
  g.reward = e->m_reward_b[b]; g.done = e->m_done_b[b];
  return slot(e, "frame_gather_cycle", 0, s, [&] { return frame_gather_multi_launch(g, n, s); });
}

int ph_frozen_batched(recnn_engine* e, int rows, int n, int run_off0, hipStream_t s) {
  const int A = e->A, nc = e->n_critic;
  const int POL = RECNN_NET_POLICY, TPOL = RECNN_NET_TARGET_POLICY;
  const int TVAL[2] = {RECNN_NET_TARGET_VALUE1, RECNN_NET_TARGET_VALUE2};
  const int actor_m1 = e->td3 ? 4 : 2;
  const int64_t aoff = (int64_t)A * 2;
  const int M = n * rows;
  int rc;
  if (e->td3 && !e->ext_noise)     // the cycle's batches in one launch (batch j: the key of step run_off0 + j)
    if ((rc = slot(e, "td3_noise", 0, s, [&] { return noise_fill_launch(e->m_noise, (int64_t)rows * A, e->hy.noise_std, e->cfg.seed, e->counters, run_off0, s, n); }))) return rc;
  const double l1_fl_a = 2.0 * M * (double)e->H * e->S, l1_fl_c = 2.0 * M * (double)e->H * (e->S + A);
  const double t_fl_a = 2.0 * M * ((double)e->H * e->H + (double)A * e->H), t_fl_c = 2.0 * M * ((double)e->H * e->H + e->H);
  if (e->tune.frozen_fused) {
    auto fill = [&](FrozenProb* p, int ni, const void* A0, int K0, int col0, int mask_idx) {
      const Net& n = e->net[ni];
      memset(p, 0, sizeof(*p));
      p->A[0] = A0; p->lda[0] = e->ldx; p->K[0] = K0; p->w1_col[0] = col0; p->nseg = 1;
      p->W1 = sh_ptr(e, ni, W1); p->ldw1 = n.ld_w1;
      p->W2 = sh_ptr(e, ni, W2); p->ldw2 = n.ld_w2;
      if (!n.critic) { p->W3 = sh_ptr(e, ni, W3); p->ldw3 = n.ld_w3; }
      p->b1 = n.p + n.off[B1]; p->b2 = n.p + n.off[B2]; p->b3 = n.p + n.off[B3];
      p->w3row = n.critic ? n.p + n.off[W3] : nullptr;
      p->rows = M; p->H = e->H; p->out_dim = n.out_dim;
      p->mask_mode = RECNN_MASK_NONE;
      if (mask_idx >= 0 && e->cfg.mask_mode == RECNN_MASK_HASH) {
        p->mask_mode = RECNN_MASK_HASH;
        p->seed = e->cfg.seed; p->stream1 = (uint32_t)mask_idx; p->stream2 = (uint32_t)mask_idx + 1;
        p->step_ptr = e->counters; p->step_add = run_off0; p->rows_per_set = rows;
      }
      p->ldh = e->Hp;
    };
    // Two launches (the target critics need the target actor's output).  A launch is 128-row workgroups at ONE per CU, so what
    // counts is how many rounds of 256 it takes: {target actor, actor} = 2 x 160 workgroups at 10 x 2048 rows = two rounds
    // with the second three quarters empty.  The actor depends on nothing here, so its batches are dealt out over both
    // launches to fill them: as many whole batches next to the target actor as fit the first round, the rest next to the
    // target critics (DDPG, 10 x 2048 rows: 160 + 96 and 160 + 64 workgroups = two full rounds instead of three).
    // Round 6: 64-row workgroups last 0.70 of a 128-row one's time (tools/frozen_trace.py: 48.5k against 69.6k clocks) -- a gain while the
    // launch still fits the same number of rounds, i.e. for the SHORT segments of a run that starts or ends inside a policy cycle (the
    // driver's 20 steps: 6 + 10 + 4).  All (rows per workgroup of launch 1, actor batches in launch 1, rows per workgroup of launch 2)
    // are priced as rounds x duration; ties go to the 128-row form and to the larger first share.
    const int cus = 256;
    int sets_a = n, fr_a = 128, fr_b = 128;
    if (rows % 128 == 0) {
      float best = 1e30f;
      for (int fa = 128; fa >= (e->tune.frozen_half ? 64 : 128); fa -= 64)
        for (int fb = 128; fb >= (e->tune.frozen_half ? 64 : 128); fb -= 64)
          for (int sa = n; sa >= 0; --sa) {
            const int wa = (n + sa) * (rows / fa), wb = (nc * n + (n - sa)) * (rows / fb);
            const float cost = (float)((wa + cus - 1) / cus) * (fa == 64 ? 0.70f : 1.0f) + (float)((wb + cus - 1) / cus) * (fb == 64 ? 0.70f : 1.0f);
            if (cost < best - 1e-6f) { best = cost; sets_a = sa; fr_a = fa; fr_b = fb; }
          }
    }
    auto actor_part = [&](FrozenProb* p, int set0, int nsets) {      // the actor on batches set0 .. set0 + nsets - 1
      const int64_t r0 = (int64_t)set0 * rows;
      fill(p, POL, e->m_xs + r0 * e->ldx * 2 + aoff, e->K1a, 0, actor_m1);
      p->rows = nsets * rows;
      p->step_add = run_off0 + set0;
      p->h1 = e->m_pa_h1 + r0 * e->Hp * 2; p->h2 = e->m_pa_h2 + r0 * e->Hp * 2;
      p->out = e->m_ga + r0 * e->Ap * 2; p->ldo = e->Ap;
    };
    FrozenBatch fb;
    int np = 0;
    fill(&fb.p[np], TPOL, e->m_xn + aoff, e->K1a, 0, -1);            // target actor on s' -> next_action into the rows' action slot
    fb.p[np].out = e->m_xn; fb.p[np].ldo = e->ldx;
    if (e->td3) { fb.p[np].addend = e->m_noise; fb.p[np].ld_add = A; fb.p[np].add_clip = e->hy.noise_clip; }
    ++np;
    if (sets_a > 0) actor_part(&fb.p[np++], 0, sets_a);              // actor on s -> gen_action, activations kept for the policy step
    if ((rc = slot(e, "frozen_actors", l1_fl_a + t_fl_a + (l1_fl_a + t_fl_a) * sets_a / n, s, [&] { return mlpf_launch(fb, np, s, fr_a); }))) return rc;
    FrozenBatch fc;
    int nq = 0;
    for (int c = 0; c < nc; ++c) {                                  // target critics on [s' | next_action]: state part first
      fill(&fc.p[nq], TVAL[c], e->m_xn + aoff, e->K1a, A, -1);
      fc.p[nq].A[1] = e->m_xn; fc.p[nq].lda[1] = e->ldx; fc.p[nq].K[1] = e->Ap; fc.p[nq].w1_col[1] = 0; fc.p[nq].nseg = 2;
      fc.p[nq].q = e->m_tq[c];
      ++nq;
    }
    if (sets_a < n) actor_part(&fc.p[nq++], sets_a, n - sets_a);
    return slot(e, "frozen_target_critics", nc * (l1_fl_c + t_fl_c) + (l1_fl_a + t_fl_a) * (n - sets_a) / n, s, [&] { return mlpf_launch(fc, nq, s, fr_b); });
  }

  {
    L1Batch lb;
    fill_l1(e, &lb.p[0], TPOL, M, e->m_xn + aoff, e->ldx, e->K1a, 0, e->m_tp_h1, -1, run_off0);
    fill_l1(e, &lb.p[1], POL, M, e->m_xs + aoff, e->ldx, e->K1a, 0, e->m_pa_h1, actor_m1, run_off0);
    lb.p[1].rows_per_set = rows;
    if ((rc = slot(e, "l1_frozen_actors", 2 * l1_fl_a, s, [&] { return l1gemm_launch(lb, 2, e->tune.l1_big, s); }))) return rc;
    if (e->tune.frozen_gemm) {
      // layers 2 and 3 of both actors as cycle-wide GEMMs through the same tiled kernel (K = 256): per output element the
      // arithmetic of the row-panel tail kernel (k ascending in steps of 32 from a zero accumulator, + bias, relu, dropout /
      // + clipped noise, round to bf16), without 1280 workgroups each starting a 192 KB weight stream for 32 rows
      const Net& tn = e->net[TPOL];
      const Net& pn = e->net[POL];
      L1Batch l2;
      fill_l1(e, &l2.p[0], TPOL, M, e->m_tp_h1, e->Hp, e->Hp, 0, e->m_tp_h2, -1, run_off0);
      l2.p[0].W1 = sh_ptr(e, TPOL, W2); l2.p[0].ldw1 = tn.ld_w2; l2.p[0].b1 = tn.p + tn.off[B2];
      fill_l1(e, &l2.p[1], POL, M, e->m_pa_h1, e->Hp, e->Hp, 0, e->m_pa_h2, actor_m1 + 1, run_off0);
      l2.p[1].W1 = sh_ptr(e, POL, W2); l2.p[1].ldw1 = pn.ld_w2; l2.p[1].b1 = pn.p + pn.off[B2];
      l2.p[1].rows_per_set = rows;
      if ((rc = slot(e, "l2_frozen_actors", 4.0 * M * (double)e->H * e->H, s, [&] { return l1gemm_launch(l2, 2, e->tune.l1_big, s); }))) return rc;
      L1Batch l3;
      fill_l1(e, &l3.p[0], TPOL, M, e->m_tp_h2, e->Hp, e->Hp, 0, e->m_xn, -1, run_off0);
      l3.p[0].W1 = sh_ptr(e, TPOL, W3); l3.p[0].ldw1 = tn.ld_w3; l3.p[0].b1 = tn.p + tn.off[B3];
      l3.p[0].H = A; l3.p[0].w_rows = e->Ap; l3.p[0].no_relu = 1; l3.p[0].ldh = e->ldx;
      if (e->td3) { l3.p[0].addend = e->ext_noise ? e->ext_noise : e->m_noise; l3.p[0].ld_add = A; l3.p[0].add_clip = e->hy.noise_clip; }
      fill_l1(e, &l3.p[1], POL, M, e->m_pa_h2, e->Hp, e->Hp, 0, e->m_ga, -1, run_off0);
      l3.p[1].W1 = sh_ptr(e, POL, W3); l3.p[1].ldw1 = pn.ld_w3; l3.p[1].b1 = pn.p + pn.off[B3];
      l3.p[1].H = A; l3.p[1].w_rows = e->Ap; l3.p[1].no_relu = 1; l3.p[1].ldh = e->Ap;
      if ((rc = slot(e, "l3_frozen_actors", 4.0 * M * (double)A * e->H, s, [&] { return l1gemm_launch(l3, 2, e->tune.l1_big, s); }))) return rc;
    } else {
      TailBatch tb;
      TailProb* p = &tb.p[0];
      fill_tail(e, p, TAIL_ACTOR, TPOL, M, e->m_tp_h1, -1, run_off0);
      p->out = e->m_xn; p->ldo = e->ldx;
      if (e->td3) { p->addend = e->ext_noise ? e->ext_noise : e->m_noise; p->ld_add = A; p->add_clip = e->hy.noise_clip; }
      p = &tb.p[1];
      fill_tail(e, p, TAIL_ACTOR, POL, M, e->m_pa_h1, actor_m1 + 1, run_off0);
      p->rows_per_set = rows;
      p->h2 = e->m_pa_h2; p->out = e->m_ga; p->ldo = e->Ap;
      if ((rc = slot(e, "tail_frozen_actors", 2 * t_fl_a, s, [&] { return mlpt_launch(tb, 2, s); }))) return rc;
    }
  }
  {
    L1Batch lb;
    for (int c = 0; c < nc; ++c) {
      fill_l1(e, &lb.p[c], TVAL[c], M, e->m_xn + aoff, e->ldx, e->K1a, A, e->m_tq_h1[c], -1, run_off0);
      l1_seg1(&lb.p[c], e->m_xn, e->ldx, e->Ap, 0);
    }
    if ((rc = slot(e, "l1_frozen_target_critic", nc * l1_fl_c, s, [&] { return l1gemm_launch(lb, nc, e->tune.l1_big, s); }))) return rc;
    if (e->tune.frozen_gemm) {
      L1Batch l2;
      for (int c = 0; c < nc; ++c) {
        const Net& t = e->net[TVAL[c]];
        fill_l1(e, &l2.p[c], TVAL[c], M, e->m_tq_h1[c], e->Hp, e->Hp, 0, c == 0 ? e->m_tp_h2 : e->m_tp_h1, -1, run_off0);   // (the target actor's panels are done with)
        l2.p[c].W1 = sh_ptr(e, TVAL[c], W2); l2.p[c].ldw1 = t.ld_w2; l2.p[c].b1 = t.p + t.off[B2];
      }
      if ((rc = slot(e, "l2_frozen_target_critic", nc * 2.0 * M * (double)e->H * e->H, s, [&] { return l1gemm_launch(l2, nc, e->tune.l1_big, s); }))) return rc;
      for (int c = 0; c < nc; ++c) {
        const Net& t = e->net[TVAL[c]];
        const void* h2 = c == 0 ? e->m_tp_h2 : e->m_tp_h1;
        if ((rc = slot(e, "q_frozen_target_critic", 2.0 * M * e->H, s, [&] { return qdot_launch(h2, e->Hp, t.p + t.off[W3], t.p + t.off[B3], e->H, M, e->m_tq[c], s); }))) return rc;
      }
    } else {
      TailBatch tb;
      for (int c = 0; c < nc; ++c) {
        fill_tail(e, &tb.p[c], TAIL_CRITIC_Q, TVAL[c], M, e->m_tq_h1[c], -1, run_off0);
        tb.p[c].q = e->m_tq[c];
      }
      if ((rc = slot(e, "tail_frozen_target_critic", nc * t_fl_c, s, [&] { return mlpt_launch(tb, nc, s); }))) return rc;
    }
  }
  return 0;
}

// The critics' exchange can run inside their optimizer launch when every workgroup's element range is made of whole float4
// groups of the arena (all tensor offsets and all but the last tensor's sizes multiples of 4) and there is a flag slot per
// workgroup.
bool comm_fused_ok(recnn_engine* e, int rows) {
  if (!e->comm || !e->comm_region || !e->tune.comm_fused) return false;
  const int VAL[2] = {RECNN_NET_VALUE1, RECNN_NET_VALUE2};
  for (int c = 0; c < e->n_critic; ++c) {
    const NetLayout L = make_layout(e, VAL[c], rows);
    if (L.nblk > COMM_MAX_WG || !L.t[W1].nslab) return false;
    for (int t = 0; t < 6; ++t) {
      if (L.t[t].p_off & 3) return false;
      if (t < 5 && (((int64_t)L.t[t].rows * L.t[t].cols) & 3)) return false;
    }
  }
  return true;
}

int net_allreduce(recnn_engine* e, int ni, const char* name, hipStream_t s) {
  Net& n = e->net[ni];
  return slot(e, name, 0, s, [&] {
    return e->comm_region ? comm_allreduce_region(e->comm, e->comm_off[ni], n.g, g_direct(e, ni) ? nullptr : n.g, n.n_params, s) : comm_allreduce_launch(e->comm, n.g, n.n_params, s);
  });
}

// The whole step.  `policy_step` is decided by the caller (host counter), everything else is on-device.
// pregathered: the batch of this step is already in the current buffer set (put there by the previous step's
// optimizer launch); gather_next: this step's critic optimizer launch also gathers the NEXT batch into the other set.
int step_impl(recnn_engine* e, int rows, bool learn, bool policy_step, hipStream_t s, bool pregathered, bool gather_next,
              bool defer_policy_fwd, bool frozen_done) {
  int rc;
  if (!pregathered && (rc = stage_batch(e, rows, s))) return rc;
  // Split bf16 with the fused dW + optimizer launch: that launch (147 KB of LDS per workgroup) cannot carry the look-ahead gather the
  // way apply_gather_kernel does, so the critic HEAD launch does -- armed here, consumed inside ph_forward (if no head launch takes it,
  // the optimizer launch carries it as before and the two-launch form stays)
  GatherArgs ga_head;
  bool head_ride = false;
  if (learn && gather_next && e->x3 && !e->comm && e->tune.dw_fuse && e->H == 256 && e->Hp == 512) {
    const NetLayout L0 = make_layout(e, RECNN_NET_VALUE1, rows);
    if (dwadam_tensor_ok(L0, W1, rows, 1) && dwadam_tensor_ok(L0, W2, rows, 1)) {
      ga_head = gather_args(e, rows, e->cur_set ^ 1, e->run_off + 1);
      e->head_gather = &ga_head;
      head_ride = true;
    }
  }
  if (frozen_done) {   // cycle mode: the batch is in place and the frozen networks have been applied to it (ph_frozen_batched)
    if ((rc = ph_forward_split(e, rows, true, true, learn, s, true))) return rc;
  } else if ((rc = ph_forward(e, rows, true, true, learn, s))) return rc;
  if (learn) {
    // The critic's soft update reads the just-updated weights and nothing reads the target before the
    // next step, so on policy steps it is fused into the critic's optimizer pass (ddpg.py:95-97) -- which itself is the
    // epilogue of the weight-gradient launch on the bf16 unit-backward path (dwopt.hip), a separate Adam launch otherwise.
    GatherArgs ga;
    const bool rode = head_ride && e->head_gather == nullptr;     // the head launch took the next step's gather
    e->head_gather = nullptr;
    if (gather_next && !rode) { ga = gather_args(e, rows, e->cur_set ^ 1, e->run_off + 1); e->pregather = &ga; }
    if (e->comm) {
      // data parallel: finished gradients into the flat arenas, summed over the ranks by one launch per arena, then the
      // replicated optimizer step on grad / world (recnn_amd/parallel.py; the arithmetic of the global batch mean)
      const int VAL[2] = {RECNN_NET_VALUE1, RECNN_NET_VALUE2};
      if (comm_fused_ok(e, rows)) {
        // ... with the exchange inside the critics' optimizer launches (optim.hip exchange_grads): the launches of the
        // single-GPU step
        rc = ph_value_backward(e, rows, false, s);
        if (!rc) rc = value_apply(e, policy_step, e->comm_scale, s, rows);
      } else {
        rc = ph_value_backward(e, rows, true, s);
        for (int c = 0; c < e->n_critic && !rc; ++c) rc = net_allreduce(e, VAL[c], "allreduce_critic", s);
        if (!rc) rc = value_apply(e, policy_step, e->comm_scale, s);
      }
    } else if ((!gather_next || rode) && dwadam_ok(e, rows)) {
      rc = ph_value_dwadam(e, rows, policy_step, s);
    } else {
      rc = ph_value_backward(e, rows, false, s);
      if (!rc) rc = value_apply(e, policy_step, 1.0f, s, rows);
    }
    e->pregather = nullptr;
    if (rc) return rc;
  }
  const bool pol = learn && policy_step;
  if (defer_policy_fwd && !pol && learn) {
    // run graphs: the next step's forward launch carries this step's policy-loss forward (see ph_forward)
    e->pending_pc.on = true; e->pending_pc.xs = e->xcs; e->pending_pc.ga = e->gen_action;
    e->pending_pc.run_off = e->run_off; e->pending_pc.slot = e->run_off;
    e->hist_pol_count[e->run_off] = rows; e->hist_pol_add[e->run_off] = 0;
  } else if ((rc = ph_policy(e, rows, pol, !e->comm, s, !learn))) {
    return rc;
  }
  if (pol && e->comm) {   // the L1 clip quirk acts on the REDUCED actor gradient
    if ((rc = net_allreduce(e, RECNN_NET_POLICY, "allreduce_actor", s))) return rc;
    if ((rc = policy_apply(e, true, e->comm_scale, s, false))) return rc;
  } else if (pol && (rc = policy_apply(e, true, 1.0f, s, true))) {
    return rc;
  }
  return ph_finish(e, rows, learn, pol, s);
}
}  // namespace recnn_eng

