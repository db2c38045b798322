// engine.hip -- the fused DDPG / TD3 update step: launch sequencing, workspace layout, hipGraph replay.
//
// Replaces the per-step Python/ATen call tree of the reference:
//   recnn/nn/update/ddpg.py:58-104, recnn/nn/update/misc.py:25-44  (DDPG)
//   recnn/nn/update/td3.py:66-150                                   (TD3)
// for Actor/Critic networks (recnn/nn/models.py:41-73, :187-213) with Adam and the soft target
// update (recnn/utils/misc.py:1-5).  ~560 ATen calls per reference step become 15 kernel launches
// (25 on a policy step), all stream-ordered, no host synchronisation, replayable as a hipGraph.
//
// Launch plan of one DDPG step (TD3 adds the twin critics to the same grouped launches):
//   G1  fwd L1  {target actor(s'), critic(s,a), actor(s)}        grouped: 3 problems, 1 launch
//   G2  fwd L2  {same three nets}
//   G3  fwd L3  {target actor -> next_action slot of xn (+TD3 noise), actor -> gen_action}
//   G4  fwd L1  {target critic([a'|s'])}      G5  fwd L2 {target critic}
//   H1  head    TD target y, Q, dQ = 2(Q-y)/B, value-loss partials
//   H2  head bwd  dz2 + partials of dW3/db3/db2
//   D1  dX      dz1 = (dz2 W2) * 2[h1>0]  + column sums (db1)
//   W1  dW      {dW2 = dz2^T h1, dW1 = dz1^T [a|s]}  split over the batch into slabs
//   R1  reduce slabs -> flat grad arena      A1  Adam (+ shadow refresh, + soft update on policy steps)
//   G6  fwd L1  critic([gen_action | s]) with the UPDATED weights (2 contraction segments)   G7 fwd L2
//   H3  head    policy loss partials
//   [policy step]  H4 head bwd, D2..D5 dX chain back to the actor, W2/W3 actor dW, R2, L1-norm, A2
//   F   loss finalize + device step counters
#include <math.h>
#include <functional>
#include <new>
#include "engine_internal.h"


// ------------------------------------------------------------------------------------ sizing
namespace recnn_eng {

struct Carver {
  char* base;
  int64_t off = 0;
  explicit Carver(char* b) : base(b) {}
  char* take(int64_t bytes) {
    off = ru(off, 256);
    char* p = base ? base + off : nullptr;
    off += bytes;
    return p;
  }
};

void net_dims(recnn_engine* e, int ni) {
  Net& n = e->net[ni];
  n.critic = ni >= RECNN_NET_VALUE1;
  n.in_dim = n.critic ? e->S + e->A : e->S;
  n.out_dim = n.critic ? 1 : e->A;
  const int H = e->H;
  n.off[W1] = 0;
  n.off[B1] = n.off[W1] + (int64_t)H * n.in_dim;
  n.off[W2] = n.off[B1] + H;
  n.off[B2] = n.off[W2] + (int64_t)H * H;
  n.off[W3] = n.off[B2] + H;
  n.off[B3] = n.off[W3] + (int64_t)n.out_dim * H;
  n.n_params = n.off[B3] + n.out_dim;
  n.ld_w1 = n.critic ? e->K1c : e->K1a;
  n.ld_w2 = e->Hp;
  n.ld_w3 = e->Hp;
  n.sh_off[W1] = 0;
  n.sh_off[W2] = (int64_t)e->Hl * n.ld_w1;
  int64_t tot = n.sh_off[W2] + (int64_t)e->Hl * n.ld_w2;
  if (!n.critic) {
    n.sh_off[W3] = tot;
    tot += (int64_t)e->Al * n.ld_w3;
  }
  n.shadow_elems = tot;
  {
    const int64_t ne[6] = {(int64_t)H * n.in_dim, H, (int64_t)H * H, H, (int64_t)n.out_dim * H, n.out_dim};
    n.n_rows_blk = 0;
    for (int i = 0; i < 6; ++i) n.n_rows_blk += opt_blocks(ne[i]);
  }
}

bool net_used(const recnn_engine* e, int ni) { return e->td3 || ni < RECNN_NET_VALUE2; }
bool net_learns(int ni) { return ni == RECNN_NET_POLICY || ni == RECNN_NET_VALUE1 || ni == RECNN_NET_VALUE2; }

// Lays out (or, with base == NULL, only measures) the workspace.
int64_t carve(recnn_engine* e, char* base) {
  Carver c(base);
  const int64_t Bc = e->Bc, Hp = e->Hp, Ap = e->Ap, H = e->H, A = e->A, es = e->esz;
  for (int ni = 0; ni < RECNN_NET_COUNT; ++ni) {
    if (!net_used(e, ni)) continue;
    Net& n = e->net[ni];
    n.shadow = c.take(n.shadow_elems * es);
  }
  auto act = [&](Acts& a) { a.h1 = c.take(Bc * Hp * es); a.h2 = c.take(Bc * Hp * es); };
  act(e->tp);
  act(e->pa);
  act(e->pc);
  for (int i = 0; i < e->n_critic; ++i) {
    act(e->tq[i]);
    act(e->cv[i]);
    e->dzc2[i] = c.take(Bc * Hp * es);
    e->dzc1[i] = c.take(Bc * Hp * es);
  }
  e->dze2 = c.take(Bc * Hp * es);
  e->dze1 = c.take(Bc * Hp * es);
  e->dag = c.take(Bc * Ap * es);
  e->dzp2 = c.take(Bc * Hp * es);
  e->dzp1 = c.take(Bc * Hp * es);
  e->gen_action0 = c.take(Bc * Ap * es);
  e->gen_action2 = c.take(Bc * Ap * es);
  e->gen_action = e->gen_action0;
  if (e->twins) {
    e->xsh = c.take(Bc * (int64_t)e->ldx * 2);
    e->xnh = c.take(Bc * (int64_t)e->ldx * 2);
  }
  if (e->twins) {
    e->xsh2 = c.take(Bc * (int64_t)e->ldx * 2);
    e->xnh2 = c.take(Bc * (int64_t)e->ldx * 2);
    e->reward2 = (float*)c.take(Bc * 4);
    e->done2 = (float*)c.take(Bc * 4);
  }
  if (e->bf16) {
    const int64_t MB = (int64_t)recnn_engine::MSET_MAX * Bc;
    for (int b = 0; b < 2; ++b) {
      e->m_xs_b[b] = c.take(MB * e->ldx * 2); e->m_xn_b[b] = c.take(MB * e->ldx * 2);
      e->m_reward_b[b] = (float*)c.take(MB * 4); e->m_done_b[b] = (float*)c.take(MB * 4);
    }
    e->m_xs = e->m_xs_b[0]; e->m_xn = e->m_xn_b[0]; e->m_reward = e->m_reward_b[0]; e->m_done = e->m_done_b[0];
    e->m_ga = c.take(MB * Ap * 2);
    e->m_tp_h1 = c.take(MB * Hp * 2); e->m_tp_h2 = c.take(MB * Hp * 2); e->m_pa_h1 = c.take(MB * Hp * 2); e->m_pa_h2 = c.take(MB * Hp * 2);
    for (int i = 0; i < e->n_critic; ++i) { e->m_tq[i] = (float*)c.take(MB * 4); e->m_tq_h1[i] = c.take(MB * Hp * 2); }
    if (e->td3) e->m_noise = (float*)c.take(MB * A * 4);
  }
  e->noise_buf = (float*)c.take(Bc * A * 4);
  e->expected = (float*)c.take(Bc * 4);
  e->target_q = (float*)c.take(Bc * 4);
  e->qpi = (float*)c.take(Bc * 4);
  e->pl_cap = (int)(2 * Bc);
  e->pl_part_base = (float*)c.take((int64_t)LOSS_HIST_MAX * e->pl_cap * 4);
  e->pl_part = e->pl_part_base;
  e->loss_ring = (float*)c.take((int64_t)LOSS_RING * 4 * 4);
  for (int i = 0; i < e->n_critic; ++i) {
    e->tc_part[i] = (float*)c.take(Bc * (int64_t)(e->Hl > 256 ? e->Hl : 256) * 4);
    e->tc_flag[i] = (int32_t*)c.take((Bc / 32 + 1) * 4);
    e->tqv[i] = (float*)c.take(Bc * 4);
    e->q_slot[i] = (float*)c.take(Bc * 4);
  }
  const int64_t nblk_head = (Bc + HEAD_ROWS_PER_BLOCK - 1) / HEAD_ROWS_PER_BLOCK;
  const int64_t nblk_hb = (Bc + HEAD_ROWS_PER_BLOCK - 1) / HEAD_ROWS_PER_BLOCK;
  const int64_t tiles_m = (Bc + 31) / 32;  // dX column-sum slabs (32 rows each)
  for (int i = 0; i < 2; ++i) {
    e->q[i] = (float*)c.take(Bc * 4);
    e->delta[i] = (float*)c.take(Bc * 4);
  }
  e->loss_part_stride = nblk_head;
  for (int i = 0; i < 3; ++i) { e->loss_part_base[i] = (float*)c.take((int64_t)LOSS_HIST_MAX * nblk_head * 4); e->loss_part[i] = e->loss_part_base[i]; }
  for (int ni = 0; ni < RECNN_NET_COUNT; ++ni) {
    if (!net_used(e, ni) || !net_learns(ni)) continue;
    Net& n = e->net[ni];
    n.gp[W1] = (float*)c.take((int64_t)SP_W1_MAX * H * n.in_dim * 4);
    n.gp[W2] = (float*)c.take((int64_t)SP_W2 * H * H * 4);
    n.gp[B1] = (float*)c.take((n.critic ? 2 : 1) * tiles_m * H * 4);   // (critics: one entry per 16 rows when the tail runs half panels)
    if (n.critic) {
      n.gp[W3] = (float*)c.take(nblk_hb * H * 4);
      n.gp[B2] = (float*)c.take(nblk_hb * H * 4);
      n.gp[B3] = (float*)c.take(nblk_hb * 4);
    } else {
      n.gp[W3] = (float*)c.take((int64_t)SP_W3 * A * H * 4);
      n.gp[B2] = (float*)c.take(tiles_m * H * 4);
      n.gp[B3] = (float*)c.take(tiles_m * A * 4);
    }
    n.l1part = (float*)c.take((int64_t)n.n_rows_blk * 4);
  }
  e->losses = (float*)c.take(32);   // float[4] losses + int32 error word of the in-launch hand-offs (losses[4], see mlp.h)
  e->coef_out = (float*)c.take(16);
  e->counters = (int32_t*)c.take(64);
  e->l1_scratch = (float*)c.take(4096);
  e->pa0 = e->pa;
  for (int i = 0; i < 2; ++i) e->tqv0[i] = e->tqv[i];
  return ru(c.off, 256);
}

int setup_dims(recnn_engine* e, const recnn_engine_config* cfg) {
  RECNN_REQUIRE(cfg, "engine: null config");
  RECNN_REQUIRE(cfg->algo == RECNN_ALGO_DDPG || cfg->algo == RECNN_ALGO_TD3, "engine: bad algo");
  RECNN_REQUIRE(cfg->dtype == RECNN_F32 || cfg->dtype == RECNN_BF16 || cfg->dtype == RECNN_BF16X3, "engine: bad dtype");
  RECNN_REQUIRE(cfg->dtype != RECNN_BF16X3 || (cfg->action_dim % 32 == 0 && cfg->hidden % 32 == 0 && cfg->action_dim <= 256 && cfg->hidden <= 256),
                "engine: the split-bf16 compute type needs action_dim and hidden to be multiples of 32, at most 256");
  RECNN_REQUIRE(cfg->state_dim > 0 && cfg->action_dim > 0 && cfg->hidden > 0 && cfg->max_rows > 0, "engine: bad dims");
  RECNN_REQUIRE(cfg->action_dim % 8 == 0 && cfg->hidden % 8 == 0, "engine: action_dim must be a multiple of 8, hidden of 8");
  e->cfg = *cfg;
  e->S = cfg->state_dim; e->A = cfg->action_dim; e->H = cfg->hidden;
  // zero-padding granularity 128 elements: whole 256-byte k stages for the bf16 LDS-DMA pipeline
  e->Hp = (int)ru(e->H, 128); e->Ap = (int)ru(e->A, 128);
  e->K1a = (int)ru(e->S, 128);
  e->K1c = (int)ru(e->S + e->A, 128);
  e->ldx = (int)ru(e->A + e->K1a, 128);
  if (e->ldx < e->K1c) e->ldx = e->K1c;
  e->Bc = (int)ru(cfg->max_rows, 64);
  e->bf16 = cfg->dtype == RECNN_BF16;
  e->x3 = cfg->dtype == RECNN_BF16X3;
  e->twins = e->bf16 || e->x3;
  e->esz = e->twins ? 2 : 4;
  e->Hl = e->Hp; e->Al = e->Ap; e->ldx32 = e->ldx;
  if (e->x3) { e->Hp *= 2; e->Ap *= 2; e->K1a *= 2; e->K1c *= 2; e->ldx *= 2; }
  e->td3 = cfg->algo == RECNN_ALGO_TD3;
  e->n_critic = e->td3 ? 2 : 1;
  for (int ni = 0; ni < RECNN_NET_COUNT; ++ni) net_dims(e, ni);
  return 0;
}

}  // namespace recnn_eng

extern "C" int recnn_engine_query(const recnn_engine_config* cfg, recnn_engine_sizes* out) {
  RECNN_REQUIRE(out, "engine_query: null out");
  recnn_engine tmp;
  int rc = setup_dims(&tmp, cfg);
  if (rc) return rc;
  out->master_floats_actor = tmp.net[RECNN_NET_POLICY].n_params;
  out->master_floats_critic = tmp.net[RECNN_NET_VALUE1].n_params;
  out->workspace_bytes = carve(&tmp, nullptr);
  out->ld_x = tmp.ldx32;
  out->x_rows = tmp.Bc;
  return 0;
}

extern "C" int recnn_engine_create(const recnn_engine_config* cfg, void* workspace, recnn_engine** out) {
  RECNN_REQUIRE(out && workspace, "engine_create: null pointer");
  RECNN_REQUIRE(((uintptr_t)workspace & 255) == 0, "engine_create: workspace must be 256-byte aligned");
  recnn_engine* e = new (std::nothrow) recnn_engine();
  RECNN_REQUIRE(e, "engine_create: out of host memory");
  int rc = setup_dims(e, cfg);
  if (rc) { delete e; return rc; }
  if ((rc = gemm_init())) { delete e; return rc; }
  if ((rc = mlp_init())) { delete e; return rc; }
  if ((rc = bwd_init())) { delete e; return rc; }
  if ((rc = l1gemm_init())) { delete e; return rc; }
  if ((rc = mlpt_init())) { delete e; return rc; }
  if ((rc = dwadam_init())) { delete e; return rc; }
  if ((rc = mlpf_init())) { delete e; return rc; }
  if ((rc = x3tail_init())) { delete e; return rc; }
  recnn_engine_tuning_init(&e->tune);
  sync_gemm_tune(e);
  e->ws = (char*)workspace;
  e->ws_bytes = carve(e, e->ws);
  for (int i = 0; i < e->n_critic; ++i) {  // hand-off flags of the chained target critics start (and rest) at 0
    rc = recnn_check_hip(hipMemset(e->tc_flag[i], 0, (size_t)(e->Bc / 32 + 1) * 4), "engine_create: flag reset");
    if (!rc) rc = recnn_check_hip(hipMemsetD32((hipDeviceptr_t)e->q_slot[i], (int)MLP_TQ_EMPTY, (size_t)e->Bc), "engine_create: slot reset");
    if (rc) { delete e; return rc; }
  }
  for (int ni = 0; ni < RECNN_NET_COUNT; ++ni) e->net[ni].t_ptr = nullptr;
  e->net[RECNN_NET_POLICY].t_ptr = e->counters + 1;
  e->net[RECNN_NET_VALUE1].t_ptr = e->counters + 2;
  e->net[RECNN_NET_VALUE2].t_ptr = e->counters + 3;
  memset(&e->hy, 0, sizeof(e->hy));
  // the pinned words loss_finalize_kernel mirrors the losses into (recnn_engine_read_losses): host-coherent, written from the device
  rc = recnn_check_hip(hipHostMalloc((void**)&e->h_stage, 16 * sizeof(float), hipHostMallocCoherent), "engine_create: pinned loss mirror");
  if (rc) { delete e; return rc; }
  memset(e->h_stage, 0, 16 * sizeof(float));
  *out = e;
  return 0;
}

namespace recnn_eng {
void drop_graphs(recnn_engine* e) {
  for (int i = 0; i < 2; ++i)
    if (e->gexec[i]) { (void)hipGraphExecDestroy(e->gexec[i]); e->gexec[i] = nullptr; }
  for (int i = 0; i <= recnn_engine::RUN_MAX; ++i) {
    if (e->grun_o[i]) { (void)hipGraphExecDestroy(e->grun_o[i]); e->grun_o[i] = nullptr; }
    if (e->grun_p[i]) { (void)hipGraphExecDestroy(e->grun_p[i]); e->grun_p[i] = nullptr; }
  }
  if (e->grun_multi) { (void)hipGraphExecDestroy(e->grun_multi); e->grun_multi = nullptr; }
  e->grun_multi_len = 0;
  for (int i = 0; i < recnn_engine::CUSTOM_MAX; ++i) {
    if (e->grun_custom[i]) { (void)hipGraphExecDestroy(e->grun_custom[i]); e->grun_custom[i] = nullptr; }
    e->grun_custom_len[i] = 0;
  }
  e->graph_rows = 0;
  for (int i = 0; i < 7; ++i)
    for (int k = 0; k < 2; ++k)
      if (e->gdp[i][k]) { (void)hipGraphExecDestroy(e->gdp[i][k]); e->gdp[i][k] = nullptr; }
}
}  // namespace recnn_eng

extern "C" void recnn_engine_destroy(recnn_engine* e) {
  if (!e) return;
  drop_graphs(e);
  if (e->h_stage) (void)hipHostFree(e->h_stage);
  delete e;
}

extern "C" int recnn_engine_bind_net(recnn_engine* e, int ni, float* params, float* grads, float* adam_m, float* adam_v) {
  RECNN_REQUIRE(e && ni >= 0 && ni < RECNN_NET_COUNT && params, "bind_net: bad arguments");
  RECNN_REQUIRE(net_used(e, ni), "bind_net: network %d is not part of this algorithm", ni);
  RECNN_REQUIRE(((uintptr_t)params & 15) == 0, "bind_net: params must be 16-byte aligned");
  Net& n = e->net[ni];
  n.p = params; n.g = grads; n.m = adam_m; n.v = adam_v;
  n.bound = true;
  return 0;
}

extern "C" int recnn_engine_bind_slow(recnn_engine* e, int ni, float* slow) {
  RECNN_REQUIRE(e && ni >= 0 && ni < RECNN_NET_COUNT && net_used(e, ni) && net_learns(ni), "bind_slow: not a learning network of this algorithm");
  RECNN_REQUIRE(!slow || ((uintptr_t)slow & 15) == 0, "bind_slow: arena must be 16-byte aligned");
  e->net[ni].slow = slow;
  drop_graphs(e);
  return 0;
}

extern "C" int recnn_engine_bind_batch(recnn_engine* e, float* xs, float* xn, float* reward, float* done) {
  RECNN_REQUIRE(e && xs && xn && reward && done, "bind_batch: null pointer");
  RECNN_REQUIRE((((uintptr_t)xs | (uintptr_t)xn) & 15) == 0, "bind_batch: packed rows must be 16-byte aligned");
  e->xs = xs; e->xn = xn; e->reward = reward; e->done = done;
  e->reward0 = reward; e->done0 = done;
  e->xcs = e->twins ? e->xsh : (char*)xs;
  e->xcn = e->twins ? e->xnh : (char*)xn;
  e->cur_set = 0;
  drop_graphs(e);
  return 0;
}

extern "C" int recnn_engine_bind_external(recnn_engine* e, const uint8_t* masks, const float* noise) {
  RECNN_REQUIRE(e, "bind_external: null engine");
  e->ext_masks = masks;
  e->ext_noise = noise;
  return 0;
}

extern "C" int recnn_engine_bind_sampler(recnn_engine* e, const recnn_sampler* m) {
  RECNN_REQUIRE(e, "bind_sampler: null engine");
  drop_graphs(e);
  if (!m) { e->has_sampler = false; return 0; }
  RECNN_REQUIRE(m->items && m->ratings && m->user_off && m->perm && m->table && m->row_off && m->cursor, "bind_sampler: null pointer");
  RECNN_REQUIRE(m->users_per_batch > 0 && m->n_batches > 0 && m->frame > 0, "bind_sampler: bad sizes");
  RECNN_REQUIRE(m->frame * m->emb_dim + m->frame == e->S && m->emb_dim == e->A,
                "bind_sampler: frame*emb+frame=%d / emb=%d do not match the engine's state_dim=%d / action_dim=%d",
                m->frame * m->emb_dim + m->frame, m->emb_dim, e->S, e->A);
  e->smp = *m;
  e->has_sampler = true;
  return 0;
}

extern "C" int recnn_engine_set_hyper(recnn_engine* e, const recnn_hyper* h) {
  RECNN_REQUIRE(e && h, "set_hyper: null pointer");
  RECNN_REQUIRE(h->policy_every > 0, "set_hyper: policy_every must be positive");
  for (int i = 0; i < 2; ++i)
    RECNN_REQUIRE(h->opt_kind[i] == RECNN_OPT_ADAM || (h->opt_kind[i] == RECNN_OPT_RANGER && h->la_k[i] >= 0), "set_hyper: bad optimizer kind");
  e->hy = *h;
  e->hyper_set = true;
  drop_graphs(e);
  return 0;
}

extern "C" int recnn_engine_set_mask_mode(recnn_engine* e, int mask_mode) {
  RECNN_REQUIRE(e && mask_mode >= RECNN_MASK_NONE && mask_mode <= RECNN_MASK_EXTERNAL, "set_mask_mode: bad arguments");
  e->cfg.mask_mode = mask_mode;
  drop_graphs(e);
  return 0;
}

// Data parallel without the host: with a connected communicator attached, every step (eager, run graphs) sums the flat gradient
// arenas over the ranks in-stream -- critics each step, actor on policy steps -- and the optimizers step on grad * grad_scale
// (1 / world).  comm == NULL detaches.  Run graphs must be (re)built afterwards.
extern "C" int recnn_engine_set_comm(recnn_engine* e, recnn_comm* comm, float grad_scale) {
  RECNN_REQUIRE(e, "set_comm: null engine");
  RECNN_REQUIRE(!comm || grad_scale > 0.f, "set_comm: grad_scale must be positive");
  drop_graphs(e);
  e->comm = comm;
  e->comm_scale = comm ? grad_scale : 1.0f;
  e->comm_region = false;
  if (comm) {
    // a region per trained network when the communicator is large enough for all of them (else every collective copies its
    // arena in and out of the one shared region)
    int64_t off = 0;
    const int nets[3] = {RECNN_NET_VALUE1, RECNN_NET_VALUE2, RECNN_NET_POLICY};
    for (int k = 0; k < 3; ++k) {
      if (!net_used(e, nets[k])) continue;
      e->comm_off[nets[k]] = off;
      off += (e->net[nets[k]].n_params + 63) & ~(int64_t)63;
    }
    e->comm_region = off <= comm_capacity(comm);
  }
  return 0;
}

extern "C" int recnn_engine_set_counters(recnn_engine* e, int policy_t, int value1_t, int value2_t, int step) {
  RECNN_REQUIRE(e, "set_counters: null engine");
  int32_t h[4] = {step, policy_t, value1_t, value2_t};
  RECNN_HIP(hipMemcpy(e->counters, h, sizeof(h), hipMemcpyHostToDevice));
  return 0;
}

// ------------------------------------------------------------------------------------ public step API
extern "C" int recnn_engine_refresh(recnn_engine* e, int ni, void* stream) {
  RECNN_REQUIRE(e && ni >= 0 && ni < RECNN_NET_COUNT && net_used(e, ni) && e->net[ni].bound, "refresh: bad network");
  return apply_net(e, ni, 0, false, 0, 1.0f, false, -1, 0.f, (hipStream_t)stream);
}

extern "C" int recnn_engine_step(recnn_engine* e, int rows, int learn, int step, void* stream) {
  int rc = check_ready(e, rows);
  if (rc) return rc;
  const bool pol = (step % e->hy.policy_every) == 0;
  use_set(e, 0);
  e->use_sampler = e->has_sampler && e->sampler_eager;
  rc = step_impl(e, rows, learn != 0, pol, (hipStream_t)stream);
  e->use_sampler = false;
  return rc;
}

extern "C" int recnn_engine_sampler_eager(recnn_engine* e, int on) {
  RECNN_REQUIRE(e, "sampler_eager: null engine");
  e->sampler_eager = on != 0;
  return 0;
}

extern "C" int recnn_engine_value_grads(recnn_engine* e, int rows, int learn, void* stream) {
  int rc = check_ready(e, rows);
  if (rc) return rc;
  use_set(e, 0);
  e->use_sampler = e->has_sampler && e->sampler_eager;
  rc = stage_batch(e, rows, (hipStream_t)stream);
  e->use_sampler = false;
  if (rc) return rc;
  if ((rc = ph_forward(e, rows, true, false, learn != 0, (hipStream_t)stream))) return rc;
  if (learn) return ph_value_backward(e, rows, true, (hipStream_t)stream);
  return 0;
}

extern "C" int recnn_engine_value_apply(recnn_engine* e, int soft, float grad_scale, void* stream) {
  RECNN_REQUIRE(e && e->hyper_set, "value_apply: engine not ready");
  return value_apply(e, soft != 0, grad_scale, (hipStream_t)stream);
}

extern "C" int recnn_engine_policy_grads(recnn_engine* e, int rows, int backward, void* stream) {
  int rc = check_ready(e, rows);
  if (rc) return rc;
  if ((rc = ph_forward(e, rows, false, true, false, (hipStream_t)stream))) return rc;
  return ph_policy(e, rows, backward != 0, false, (hipStream_t)stream, false);
}

extern "C" int recnn_engine_policy_apply(recnn_engine* e, int soft, float grad_scale, void* stream) {
  RECNN_REQUIRE(e && e->hyper_set, "policy_apply: engine not ready");
  return policy_apply(e, soft != 0, grad_scale, (hipStream_t)stream);
}

extern "C" int recnn_engine_clip_policy_grads(recnn_engine* e, float grad_scale, void* stream) {
  RECNN_REQUIRE(e, "clip_policy_grads: null engine");
  int rc = ph_policy_l1(e, (hipStream_t)stream);
  if (rc) return rc;
  Net& pn = e->net[RECNN_NET_POLICY];
  NetLayout L = make_layout(e, RECNN_NET_POLICY, 0);
  return scale_grads_launch(L, pn.g, pn.l1part, L.nblk, grad_scale, (hipStream_t)stream);
}

extern "C" int recnn_engine_soft_update(recnn_engine* e, int ni, int target_ni, float tau, void* stream) {
  RECNN_REQUIRE(e && ni >= 0 && ni < RECNN_NET_COUNT && target_ni >= 0 && target_ni < RECNN_NET_COUNT, "soft_update: bad nets");
  RECNN_REQUIRE(net_used(e, ni) && net_used(e, target_ni) && e->net[ni].bound && e->net[target_ni].bound, "soft_update: unbound");
  return apply_net(e, ni, 0, false, 0, 1.0f, false, target_ni, tau, (hipStream_t)stream);
}

extern "C" int recnn_engine_finish(recnn_engine* e, int rows, int value_stepped, int policy_stepped, void* stream) {
  RECNN_REQUIRE(e && rows > 0, "finish: bad arguments");
  e->use_sampler = e->has_sampler && e->sampler_eager;   // the step's batch came from the sampler: advance its cursor
  const int rc = ph_finish(e, rows, value_stepped != 0, policy_stepped != 0, (hipStream_t)stream);
  e->use_sampler = false;
  return rc;
}

// A non-zero error word means a cross-workgroup wait inside a launch ran out (mlp.h): the step's numbers are void.
static int report_handoff_error(recnn_engine* e, int32_t word, hipStream_t s) {
  if (!word) return 0;
  (void)hipMemsetAsync(e->losses + 4, 0, sizeof(int32_t), s);
  (void)hipStreamSynchronize(s);
  recnn_set_error("engine: an in-launch hand-off timed out (error word 0x%x:%s%s) -- the launch order / residency contract of "
                  "the fused forward was violated; the results of the steps since the last read are invalid", word,
                  (word & MLP_ERR_PART_TIMEOUT) ? " target-critic layer-1 part" : "", (word & MLP_ERR_Q_TIMEOUT) ? " Q(s,a) slot" : "");
  return RECNN_E_STATE;
}

extern "C" int recnn_engine_read_counters(recnn_engine* e, int32_t* h_out, void* stream) {
  RECNN_REQUIRE(e && h_out, "read_counters: null pointer");
  int32_t word = 0;
  RECNN_HIP(hipMemcpyAsync(h_out, e->counters, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, (hipStream_t)stream));
  RECNN_HIP(hipMemcpyAsync(&word, e->losses + 4, sizeof(int32_t), hipMemcpyDeviceToHost, (hipStream_t)stream));
  RECNN_HIP(hipStreamSynchronize((hipStream_t)stream));
  return report_handoff_error(e, word, (hipStream_t)stream);
}

extern "C" int recnn_engine_read_losses(recnn_engine* e, float* h_out, void* stream) {
  RECNN_REQUIRE(e && h_out, "read_losses: null pointer");
  // loss_finalize_kernel -- the last launch of every step and run graph -- writes the losses and the error word into the pinned mirror
  // itself: no device-to-host copy behind the graph (that copy was a blit kernel + a second boundary: 16 us per read, round 6)
  RECNN_HIP(hipStreamSynchronize((hipStream_t)stream));
  volatile float* h = e->h_stage;
  for (int i = 0; i < 4; ++i) h_out[i] = h[i];
  const int32_t word = ((volatile int32_t*)e->h_stage)[4];
  if (word) ((volatile int32_t*)e->h_stage)[4] = 0;
  return report_handoff_error(e, word, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------ debug buffers
extern "C" int recnn_engine_unit_backward(recnn_engine* e) { return e && e->unit_bwd ? 1 : 0; }

extern "C" const void* recnn_engine_buffer(recnn_engine* e, const char* name, int64_t* rows, int64_t* cols, int64_t* ld,
                                           int* is_f32) {
  if (!e || !name) return nullptr;
  struct Ent { const char* n; const void* p; int64_t c, l; int f; };
  const int64_t Hp = e->Hp;
  const Ent tab[] = {
      {"next_action", e->xcn, e->A, e->ldx, e->twins ? 0 : 1}, {"gen_action", e->gen_action, e->A, e->Ap, 0},
      {"expected", e->expected, 1, 1, 1},          {"target_q", e->target_q, 1, 1, 1},
      {"q1", e->q[0], 1, 1, 1},                    {"q2", e->q[1], 1, 1, 1},
      {"delta1", e->delta[0], 1, 1, 1},            {"delta2", e->delta[1], 1, 1, 1},
      {"q_pi", e->qpi, 1, 1, 1},                   {"losses", e->losses, 4, 4, 1},
      {"clip_coef", e->coef_out, 1, 1, 1},         {"loss_ring", e->loss_ring, 4, 4, 1},
      {"critic1_h1", e->cv[0].h1, e->H, Hp, 0},    {"critic1_h2", e->cv[0].h2, e->H, Hp, 0},
      {"actor_h1", e->pa.h1, e->H, Hp, 0},         {"actor_h2", e->pa.h2, e->H, Hp, 0},
      {"critic1_dz2", e->dzc2[0], e->H, Hp, 0},    {"critic1_dz1", e->dzc1[0], e->H, Hp, 0},
      {"dact", e->dag, e->A, e->Ap, 0},
      {"pc_h1", e->pc.h1, e->H, Hp, 0},            {"pc_h2", e->pc.h2, e->H, Hp, 0},
      {"dze2", e->dze2, e->H, Hp, 0},              {"dze1", e->dze1, e->H, Hp, 0},
      {"dzp2", e->dzp2, e->H, Hp, 0},              {"dzp1", e->dzp1, e->H, Hp, 0},            {"noise", e->noise_buf, e->A, e->A, 1},
  };
  for (const Ent& t : tab)
    if (!strcmp(t.n, name)) {
      if (rows) *rows = (!strcmp(name, "losses") || !strcmp(name, "clip_coef")) ? 1 : (!strcmp(name, "loss_ring") ? LOSS_RING : e->cfg.max_rows);
      if (cols) *cols = t.c;
      if (ld) *ld = t.l;
      if (is_f32) *is_f32 = t.f;
      return t.p;
    }
  return nullptr;
}


