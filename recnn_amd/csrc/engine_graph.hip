// engine_graph.hip -- the step engine's profile mode and hipGraph side (split out of engine.hip in round 5): per-launch profile,
// capture of the run-graph family / made-to-order graphs / cycle-mode segments, graph replay, data-parallel phase graphs.
// The launch plans it captures are engine.hip's (engine_internal.h).
#include "engine_internal.h"

// ------------------------------------------------------------------------------------ per-launch profile
extern "C" int recnn_engine_profile(recnn_engine* e, int rows, int policy_steps, int n_steps, void* stream, float* h_ms,
                                    double* h_flops, const char** h_names, int* h_n) {
  int rc = check_ready(e, rows);
  if (rc) return rc;
  // idempotent kernels run 8x back to back inside their event pair: amortises the ~3 us an event pair costs
  e->prof_repeat = 8;
  RECNN_REQUIRE(h_ms && h_flops && h_names && h_n && n_steps > 0, "profile: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  if (!e->prof_ready) {
    for (int i = 0; i < 2 * recnn_engine::PROF_MAX; ++i) RECNN_HIP(hipEventCreate(&e->prof_ev[i]));
    e->prof_ready = true;
  }
  double acc[recnn_engine::PROF_MAX];
  for (int i = 0; i < recnn_engine::PROF_MAX; ++i) acc[i] = 0.0;
  int nslots = 0;
  for (int it = 0; it < n_steps; ++it) {
    e->prof_on = true;
    e->prof_n = 0;
    e->use_sampler = e->has_sampler;
    if (!rc && policy_steps == 2) {
      // cycle mode, as the long run graphs issue it: one policy cycle's batches, the frozen networks on all of them, then
      // the first step of the cycle (an ordinary step) on the split forward
      RECNN_REQUIRE(cycle_ok(e, rows), "profile: cycle mode is not available for this engine / batch size");
      const int n = e->hy.policy_every < recnn_engine::MSET_MAX ? e->hy.policy_every : recnn_engine::MSET_MAX;
      select_mbuf(e, 0);
      rc = ph_gather_cycle(e, rows, n, 0, 0, s);
      if (!rc) rc = ph_frozen_batched(e, rows, n, 0, s);
      if (!rc) {
        use_mset(e, 0, rows);
        rc = step_impl(e, rows, true, false, s, true, false, false, true);
      }
      leave_mset(e);
    } else if (!rc) {
      rc = step_impl(e, rows, true, policy_steps != 0, s);
    }
    e->use_sampler = false;
    e->prof_on = false;
    if (rc) return rc;
    RECNN_HIP(hipStreamSynchronize(s));
    nslots = e->prof_n;
    for (int i = 0; i < nslots; ++i) {
      float ms = 0.f;
      RECNN_HIP(hipEventElapsedTime(&ms, e->prof_ev[2 * i], e->prof_ev[2 * i + 1]));
      acc[i] += ms / e->prof_reps[i];
    }
  }
  const int cap = *h_n;
  const int n = nslots < cap ? nslots : cap;
  for (int i = 0; i < n; ++i) {
    h_ms[i] = (float)(acc[i] / n_steps);
    h_flops[i] = e->prof_flops[i];
    h_names[i] = e->prof_name[i];
  }
  *h_n = n;
  return 0;
}

// ------------------------------------------------------------------------------------ graphs
// tuning.graph_run: steps per run graph; -1 = whole policy cycles, up to 64 steps (policy_every > 32: 16 ordinary steps); 0 / 1 = off

// Executable graphs: one ordinary step, one policy step, and a family of RUN graphs (several consecutive steps per
// graph launch) -- between two graph launches the GPU idles for ~8 us (rocprofv3 kernel trace), inside a graph the
// kernels are back to back, the sampler + gather of step t+1 rides on step t's optimizer launch and the policy-loss
// forward of step t on step t+1's forward launch.  Which steps of a run are policy steps is frozen at capture time, so
// the family holds every shape graph_run() needs to cover ANY (first_step, n_steps):
//   grun_o[k]   k ordinary steps                  -- the stretch up to the next policy step / the end of the request
//   grun_p[k]   a policy step + k ordinary steps  -- k = policy_every-1 is a whole cycle, smaller k the request's tail
//   grun_multi  as many whole cycles as fit 64 steps
// (policy_every <= 17: every k; larger: k in {1, 2, 4, 8, 16, 32} and greedy composition.)
namespace recnn_eng {
// One run of `len` steps captured into *out.  phase = (number of the run's first step) mod policy_every: step i of the run is
// a policy step when (phase + i) is a multiple of policy_every; phase < 0: no policy step in the run.
int capture_run(recnn_engine* e, int rows, hipStream_t s, int phase, int len, hipGraphExec_t* out) {
  if (*out) { (void)hipGraphExecDestroy(*out); *out = nullptr; }
  const int pe = e->hy.policy_every;
  const bool look = lookahead_ok(e) && len > 1;
  hipGraph_t graph = nullptr;
  int rc = 0;
  RECNN_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  e->use_sampler = e->has_sampler;
  int n_pol = 0;
  const bool cyc = !rc && len > 1 && (e->tune.split_fwd >= 2 || len >= e->tune.cycle_min_len) && cycle_ok(e, rows);
  auto is_pol = [&](int i) { return phase >= 0 && ((phase + i) % pe) == 0; };
  if (cyc) {
    // segments = the steps up to and including the next policy step (the frozen networks change right after it)
    int seg0[recnn_engine::RUN_MAX + 1], seg1[recnn_engine::RUN_MAX + 1], nseg = 0;
    for (int i0 = 0; i0 < len;) {
      int i1 = i0;
      while (i1 + 1 < len && !is_pol(i1) && i1 - i0 + 1 < recnn_engine::MSET_MAX) ++i1;
      seg0[nseg] = i0; seg1[nseg] = i1; ++nseg;
      i0 = i1 + 1;
    }
    rc = ph_gather_cycle(e, rows, seg1[0] - seg0[0] + 1, seg0[0], 0, s);
    for (int k = 0; k < nseg && !rc; ++k) {
      const int i0 = seg0[k], i1 = seg1[k], n = i1 - i0 + 1, buf = k & 1;
      select_mbuf(e, buf);
      e->run_off = i0;
      // a one- or two-step segment (a graph that starts ON a policy step has one at its head) does not pay for the batched
      // launches (two 128-row-panel launches cost ~60 us whatever n is): its steps run the fused forward on the cycle arrays.
      // (The driver's 20-step request in cycle mode, segments 6 + 10 + 4, with this threshold at 3 / 5 / 8: 67.7 / 67.1 / 67.6
      // us/step against 67.8-68.3 all-fused -- inside the noise, so short graphs stayed on the fused schedule: cycle_min_len 30.
      //  Round 5, after the kernel-argument prefetches: the same command 67.6-68.2 in cycle mode against 69.3-69.4 fused, A/B inside one
      //  call -- cycle_min_len is 20 now.)
      const bool batched = n >= e->tune.cycle_min_seg;
      if (batched) rc = ph_frozen_batched(e, rows, n, i0, s);
      for (int i = i0; i <= i1 && !rc; ++i) {
        const bool pol = is_pol(i);
        use_mset(e, i - i0, rows);
        e->run_off = i;
        use_hist_slot(e, i < LOSS_HIST_MAX ? i : 0);
        e->hist_pol_count[i < LOSS_HIST_MAX ? i : 0] = 0;
        for (int ni = 0; ni < RECNN_NET_COUNT; ++ni) e->run_t_off[ni] = (ni == RECNN_NET_POLICY) ? n_pol : i;
        e->run_skip_finish = i + 1 < len;
        if (pol) ++n_pol;
        e->run_tick[0] = len; e->run_tick[1] = len; e->run_tick[2] = n_pol;
        // the policy-loss forward of an ordinary step rides on the next step's critic launches (its batch must survive until
        // then: not across a segment boundary)
        const bool defer = e->tune.defer_policy_fwd && i < i1;
        rc = step_impl(e, rows, true, pol, s, true, false, defer, batched);
      }
      // (the next segment's batches go into the other copy of the cycle arrays: this segment's deferred forwards still read theirs)
      if (!rc && k + 1 < nseg) rc = ph_gather_cycle(e, rows, seg1[k + 1] - seg0[k + 1] + 1, seg0[k + 1], buf ^ 1, s);
    }
    select_mbuf(e, 0);
  }
  if (cyc) leave_mset(e);
  for (int i = 0; !cyc && i < len && !rc; ++i) {
    const bool pol = phase >= 0 && ((phase + i) % pe) == 0;
    use_set(e, look ? (i & 1) : 0);
    // counters are ticked once, by the last step's finalize: step i runs `i` steps ahead of them
    e->run_off = i;
    use_hist_slot(e, i < LOSS_HIST_MAX ? i : 0);
    e->hist_pol_count[i < LOSS_HIST_MAX ? i : 0] = 0;
    for (int ni = 0; ni < RECNN_NET_COUNT; ++ni) e->run_t_off[ni] = (ni == RECNN_NET_POLICY) ? n_pol : i;
    e->run_skip_finish = i + 1 < len;
    if (pol) ++n_pol;
    e->run_tick[0] = len; e->run_tick[1] = len; e->run_tick[2] = n_pol;
    // the policy-loss forward of an ordinary step rides on the next step's forward launch (needs the second
    // buffer set)
    const bool defer = look && e->tune.defer_policy_fwd && (value_chain_ok(e) || e->x3) && i + 1 < len;
    rc = step_impl(e, rows, true, pol, s, look && i > 0, look && i + 1 < len, defer);
  }
  e->run_off = 0;
  use_hist_slot(e, 0);
  e->pending_pc.on = false;
  for (int ni = 0; ni < RECNN_NET_COUNT; ++ni) e->run_t_off[ni] = 0;
  e->run_skip_finish = false;
  e->run_tick[0] = e->run_tick[1] = e->run_tick[2] = 1;
  e->use_sampler = false;
  use_set(e, 0);
  hipError_t ce = hipStreamEndCapture(s, &graph);
  if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
  RECNN_HIP(ce);
  hipError_t ie = hipGraphInstantiate(out, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  RECNN_HIP(ie);
  // pay the one-time device-side set-up of the executable graph now, not inside somebody's timed first replay
  if (hipGraphUpload(*out, s) != hipSuccess) (void)hipGetLastError();
  return 0;
}
// the run lengths kept for `limit` (largest useful length): every length when the family stays small, else powers of two
bool run_len_kept(int k, int limit) { return k >= 1 && k <= limit && (limit <= 16 || (k & (k - 1)) == 0); }
}  // namespace recnn_eng

extern "C" int recnn_engine_graph_build(recnn_engine* e, int rows, void* stream) {
  int rc = check_ready(e, rows);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  RECNN_REQUIRE(s != nullptr, "graph_build: capture needs a non-null stream");
  drop_graphs(e);
  const int pe = e->hy.policy_every;
  int cap = e->tune.graph_run < 0 ? recnn_engine::RUN_MAX : e->tune.graph_run;   // longest run graph wanted
  if (cap > recnn_engine::RUN_MAX) cap = recnn_engine::RUN_MAX;
  if ((rc = capture_run(e, rows, s, -1, 1, &e->gexec[0]))) return rc;
  if ((rc = capture_run(e, rows, s, 0, 1, &e->gexec[1]))) return rc;
  if (cap >= 2) {
    const int o_max = pe - 1 < cap ? pe - 1 : cap;            // ordinary stretch: never across a policy step
    for (int k = 2; k <= o_max; ++k)
      if (run_len_kept(k, o_max) && (rc = capture_run(e, rows, s, -1, k, &e->grun_o[k]))) return rc;
    const int p_max = pe - 1 < cap - 1 ? pe - 1 : cap - 1;    // policy step + k ordinary ones
    for (int k = 1; k <= p_max; ++k)
      if (run_len_kept(k, p_max) && (rc = capture_run(e, rows, s, 0, k + 1, &e->grun_p[k]))) return rc;
    const int cycles = cap / pe;
    if (cycles >= 2) {
      if ((rc = capture_run(e, rows, s, 0, cycles * pe, &e->grun_multi))) return rc;
      e->grun_multi_len = cycles * pe;
    }
  }
  e->grun_look = lookahead_ok(e);
  e->graph_rows = rows;
  return 0;
}

// A run graph made to order for requests of exactly n_steps steps starting at a step congruent to first_step modulo
// policy_every: graph_run() then serves such a request with ONE launch instead of the [stretch][cycles][tail] composition
// (each graph boundary costs ~45 us of GPU time plus a host launch).  Up to CUSTOM_MAX are kept, oldest replaced.
extern "C" int recnn_engine_graph_prepare(recnn_engine* e, int first_step, int n_steps, void* stream) {
  RECNN_REQUIRE(e && e->gexec[0] && e->gexec[1], "graph_prepare: build the graphs first (recnn_engine_graph_build)");
  RECNN_REQUIRE(first_step >= 0 && n_steps >= 2 && n_steps <= recnn_engine::RUN_MAX, "graph_prepare: 2 <= n_steps <= %d",
                recnn_engine::RUN_MAX);
  hipStream_t s = (hipStream_t)stream;
  RECNN_REQUIRE(s != nullptr, "graph_prepare: capture needs a non-null stream");
  const int phase = first_step % e->hy.policy_every;
  for (int i = 0; i < recnn_engine::CUSTOM_MAX; ++i)
    if (e->grun_custom[i] && e->grun_custom_phase[i] == phase && e->grun_custom_len[i] == n_steps) return 0;
  const int slot = e->grun_custom_next;
  e->grun_custom_next = (slot + 1) % recnn_engine::CUSTOM_MAX;
  e->grun_custom_len[slot] = 0;
  const int rc = capture_run(e, e->graph_rows, s, phase, n_steps, &e->grun_custom[slot]);
  if (rc) return rc;
  e->grun_custom_phase[slot] = phase;
  e->grun_custom_len[slot] = n_steps;
  return 0;
}

// Replays n_steps consecutive steps starting at step number first_step with as few graph launches as the family allows:
// [ordinary stretch up to the next policy step] [multi-cycle graphs] [whole cycles] [policy step + tail].
extern "C" int recnn_engine_graph_run(recnn_engine* e, int first_step, int n_steps, void* stream) {
  RECNN_REQUIRE(e && e->gexec[0] && e->gexec[1], "graph_run: graphs not built");
  const int pe = e->hy.policy_every;
  hipStream_t s = (hipStream_t)stream;
  for (int c = 0; c < recnn_engine::CUSTOM_MAX; ++c)
    if (e->grun_custom[c] && e->grun_custom_len[c] == n_steps && e->grun_custom_phase[c] == first_step % pe) {
      RECNN_HIP(hipGraphLaunch(e->grun_custom[c], s));
      use_set(e, e->grun_look ? ((n_steps - 1) & 1) : 0);
      return 0;
    }
  int i = 0;
  while (i < n_steps) {
    const int step = first_step + i, rem = n_steps - i;
    const bool pol = (step % pe) == 0;
    hipGraphExec_t g = nullptr;
    int len = 1;
    if (pol) {
      if (e->grun_multi && rem >= e->grun_multi_len) { g = e->grun_multi; len = e->grun_multi_len; }
      else {
        int k = (rem < pe ? rem : pe) - 1;                    // ordinary steps that may follow inside this cycle
        if (k > recnn_engine::RUN_MAX) k = recnn_engine::RUN_MAX;
        while (k >= 1 && !e->grun_p[k]) --k;
        if (k >= 1) { g = e->grun_p[k]; len = k + 1; } else g = e->gexec[1];
      }
    } else {
      int k = pe - (step % pe);                               // ordinary steps before the next policy step
      if (k > rem) k = rem;
      if (k > recnn_engine::RUN_MAX) k = recnn_engine::RUN_MAX;
      while (k >= 2 && !e->grun_o[k]) --k;
      if (k >= 2) { g = e->grun_o[k]; len = k; } else g = e->gexec[0];
    }
    RECNN_HIP(hipGraphLaunch(g, s));
    use_set(e, (e->grun_look && len > 1) ? ((len - 1) & 1) : 0);   // where the debug views find the last batch
    i += len;
  }
  return 0;
}

// Data-parallel phase graphs (see recnn_amd/parallel.py): the gradient all-reduces run between them.
//   kind 0 H    batch + all forwards + critic backward + slab reduction            -> all-reduce critic grads
//   kind 1 T1   [ordinary step] critic Adam, policy loss, finish
//   kind 2 T2   [policy step]   critic Adam (+soft), policy loss + actor backward    -> all-reduce actor grads
//   kind 3 T3   [policy step]   L1 clip + actor Adam (+soft), finish
//   kind 4      (overlap mode)  actor forward alone: runs while the critic all-reduce is in flight; graph 0 then
//                               leaves the actor out of its first group
//   kind 5 T1H  T1 of step t followed by H of step t+1 in ONE graph (one graph launch per step instead of two: the GPU
//               idles ~8 us between graph launches); with two batch buffer sets the sampler + gather of step t+1 rides
//               on step t's critic optimizer launch, as in the single-GPU run graphs
//   kind 6 T3H  T3 followed by H of the next step
// `which` of recnn_engine_dp_graph_launch = kind + 8 * set (the buffer set step t's batch is in).
static int dp_capture(recnn_engine* e, hipStream_t s, hipGraphExec_t* out, const std::function<int()>& body) {
  if (*out) { (void)hipGraphExecDestroy(*out); *out = nullptr; }
  hipGraph_t graph = nullptr;
  RECNN_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  const int rc = body();
  hipError_t ce = hipStreamEndCapture(s, &graph);
  if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
  RECNN_HIP(ce);
  hipError_t ie = hipGraphInstantiate(out, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  RECNN_HIP(ie);
  (void)e;
  return 0;
}

extern "C" int recnn_engine_dp_graph_build(recnn_engine* e, int rows, float grad_scale, int overlap_actor, void* stream) {
  int rc = check_ready(e, rows);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  RECNN_REQUIRE(s != nullptr, "dp_graph_build: capture needs a non-null stream");
  for (int i = 0; i < 7; ++i)
    for (int k = 0; k < 2; ++k)
      if (e->gdp[i][k]) { (void)hipGraphExecDestroy(e->gdp[i][k]); e->gdp[i][k] = nullptr; }
  const bool look = !overlap_actor && lookahead_ok(e);
  e->dp_sets = look ? 2 : 1;
  struct SamplerScope { recnn_engine* e; ~SamplerScope() { e->use_sampler = false; } } sampler_scope{e};
  e->use_sampler = e->has_sampler;
  auto head = [&](bool pregathered) -> int {
    int r = pregathered ? 0 : stage_batch(e, rows, s);
    if (!r && !(r = ph_forward(e, rows, true, !overlap_actor, true, s))) r = ph_value_backward(e, rows, true, s);
    return r;
  };
  use_set(e, 0);
  if ((rc = dp_capture(e, s, &e->gdp[0][0], [&] { return head(false); }))) return rc;
  if (overlap_actor && (rc = dp_capture(e, s, &e->gdp[4][0], [&] { return ph_forward(e, rows, false, true, false, s); }))) return rc;
  for (int set = 0; set < e->dp_sets && !rc; ++set) {
    use_set(e, set);
    use_hist_slot(e, set);   // loss partial sums of a step live in the slot of its batch buffer set
    rc = dp_capture(e, s, &e->gdp[1][set], [&] {
      int r;
      if (!(r = value_apply(e, false, grad_scale, s)) && !(r = ph_policy(e, rows, false, false, s, false))) r = ph_finish(e, rows, true, false, s);
      return r;
    });
    if (!rc) rc = dp_capture(e, s, &e->gdp[2][set], [&] {
      int r;
      if (!(r = value_apply(e, true, grad_scale, s))) r = ph_policy(e, rows, true, false, s, false);
      return r;
    });
    const int pol_dots = e->pl_dot_parts;   // how the policy step's loss partials were produced (by graph 2's ph_policy)
    if (!rc) rc = dp_capture(e, s, &e->gdp[3][set], [&] {
      int r;
      e->pl_dot_parts = pol_dots;
      if (!(r = policy_apply(e, true, grad_scale, s))) r = ph_finish(e, rows, true, true, s);
      return r;
    });
    if (overlap_actor) continue;   // the overlap variant keeps one graph per phase
    if (!rc) rc = dp_capture(e, s, &e->gdp[5][set], [&] {
      GatherArgs ga;
      if (look) { ga = gather_args(e, rows, set ^ 1, 1); e->pregather = &ga; }
      int r = value_apply(e, false, grad_scale, s);
      e->pregather = nullptr;
      // With two buffer sets the policy-loss forward of step t rides on step t+1's forward launch (as in the run
      // graphs); step t's finalize then closes the graph: the head of step t+1 is captured one step ahead of the
      // device counters and writes its loss partial sums into the other per-step slot.
      const bool defer = look && e->tune.defer_policy_fwd && value_chain_ok(e);
      if (!r) {
        if (defer) {
          e->pending_pc.on = true; e->pending_pc.xs = e->xcs; e->pending_pc.ga = e->gen_action; e->pending_pc.run_off = 0; e->pending_pc.slot = set;
        } else if (!(r = ph_policy(e, rows, false, false, s, false))) {
          r = ph_finish(e, rows, true, false, s);
        }
      }
      if (!r) {
        if (look) use_set(e, set ^ 1);
        if (defer) { e->run_off = 1; use_hist_slot(e, set ^ 1); }
        r = head(look);
        if (defer) {
          e->run_off = 0;
          use_hist_slot(e, set);
          e->pl_dot_parts = rows;   // Q per row (b3 included) in this step's policy slot
          if (!r) r = ph_finish(e, rows, true, false, s);
          e->pl_dot_parts = 0;
        }
        e->pending_pc.on = false;
        use_set(e, set);
      }
      return r;
    });
    if (!rc) rc = dp_capture(e, s, &e->gdp[6][set], [&] {
      int r;
      e->pl_dot_parts = pol_dots;
      if (!(r = policy_apply(e, true, grad_scale, s))) r = ph_finish(e, rows, true, true, s);
      if (!r) {
        if (look) { use_set(e, set ^ 1); use_hist_slot(e, set ^ 1); }
        r = head(false);   // the policy step's tail does not look ahead: its own gather, after the cursor tick
        use_set(e, set);
        use_hist_slot(e, set);
      }
      return r;
    });
  }
  use_set(e, 0);
  use_hist_slot(e, 0);
  return rc;
}

extern "C" int recnn_engine_dp_sets(recnn_engine* e) { return e ? e->dp_sets : 1; }

extern "C" int recnn_engine_dp_graph_launch(recnn_engine* e, int which, void* stream) {
  const int kind = which & 7, set = which >> 3;
  RECNN_REQUIRE(e && which >= 0 && kind < 7 && set < 2 && e->gdp[kind][set], "dp_graph_launch: graph %d not built", which);
  RECNN_HIP(hipGraphLaunch(e->gdp[kind][set], (hipStream_t)stream));
  // where the debug views find the batch afterwards: merged graphs (kinds 5, 6) end in the other set's head
  use_set(e, (kind >= 5 && e->dp_sets == 2) ? (set ^ 1) : set);
  return 0;
}

