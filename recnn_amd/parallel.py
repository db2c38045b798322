"""Data-parallel replicas of the update step over RCCL (new functionality; the reference has no distributed code).

Weak scaling, one process per GPU: every rank owns a shard of the replay users (rank r takes perm[r::W]), builds
its own batch and runs the forward/backward phases locally; the only exchange is one all-reduce(sum) of the flat
fp32 gradient arena per optimizer step -- critic(s) every step (1.7 MB each), actor every `policy_step`-th step.
The L1 clip quirk is evaluated on the REDUCED actor gradient, optimizer / soft update run replicated, so all
ranks hold identical weights without any broadcast (`check_replicas` verifies).

N ranks x B/N rows with the matching slices of the dropout masks == 1 rank x B rows up to fp32 summation order
(tests/test_parallel_gloo.py proves the orchestration with the CPU oracle standing in for the engine).
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist

from . import _lib as L

__all__ = ["DataParallelStepper", "PeerComm", "VocabParallelCritic", "VocabParallelDiscreteActor", "shard_users"]


def shard_users(perm: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    """Rank r's share of an epoch permutation of user slots."""
    return perm[rank::world].contiguous()


class PeerComm:
    """The device-side gradient exchange `recnn_dp_allreduce_flat` (csrc/comm.hip): one peer-mapped buffer per rank (hipIpc;
    one process per GPU of ONE node, world <= 8), a two-shot all-reduce as a single kernel launch -- a graph node.

    The handles travel over the given torch.distributed group once, at construction (`all_gather_object`); afterwards the
    group is not used again.  `PeerComm.create` returns None on every rank when any rank cannot map its peers (another node,
    IPC unavailable): the caller then stays on RCCL (`dist.all_reduce`)."""

    def __init__(self, max_floats: int, group=None, _state=None, memory_kind=None, workgroups=None):
        self.group = group
        self.memory_kind, self.workgroups = memory_kind, workgroups
        ready = dist.is_initialized()
        self.world = dist.get_world_size(group) if ready else 1
        self.rank = dist.get_rank(group) if ready else 0
        self.max_floats = int(max_floats)
        self.handle = None
        try:
            mine, err = 0, None
            try:
                self._open()
                if self.world > 1:
                    mine = self._export()
            except L.RecnnHipError as ex:
                if self.world == 1:
                    raise
                err = ex
            if self.world > 1:
                # every rank contributes its handle, or a non-handle when it could not create / export its buffer: this gather is
                # the agreement -- EVERY rank takes part in it exactly once (`_state` tells `create` that this rank has), and nobody
                # connects to a partial set
                every = [None] * self.world
                dist.all_gather_object(every, mine, group=group)
                if _state is not None:
                    _state["exchanged"] = True
                if err is not None:
                    raise err
                if any(not isinstance(h, (bytes, bytearray)) or len(h) != len(mine) for h in every):
                    raise L.RecnnHipError("PeerComm: a rank could not export its peer buffer")
                self._connect(every)
        except Exception:
            self.close()
            raise

    # ---- the three library steps of the constructor (tests/test_peercomm_vote_gloo.py replaces them to provoke failures)
    def _open(self):
        import ctypes as C
        self.lib = L.load()
        h = C.c_void_p()
        # settings of THIS communicator (the library keeps no process-wide communicator state): memory kind of the peer buffers and
        # workgroups per collective, from the constructor's keywords or, for A/B runs from the shell, the environment
        mem = self.memory_kind if self.memory_kind is not None else int(os.environ.get("RECNN_COMM_MEMORY", "0") or 0)
        L.call("recnn_comm_create_ex", self.world, self.rank, self.max_floats, mem, C.byref(h))
        self.handle = h
        wg = self.workgroups if self.workgroups is not None else os.environ.get("RECNN_COMM_WORKGROUPS")
        if wg not in (None, ""):
            L.call("recnn_comm_set_workgroups", self.handle, int(wg))

    def _export(self) -> bytes:
        import ctypes as C
        nb = int(self.lib.recnn_comm_handle_bytes())
        mine = C.create_string_buffer(nb)
        L.call("recnn_comm_export", self.handle, mine, nb)
        return bytes(mine.raw)

    def _connect(self, every):
        import ctypes as C
        nb = len(every[0])
        blob = C.create_string_buffer(b"".join(every), nb * self.world)
        L.call("recnn_comm_connect", self.handle, blob, nb)

    @staticmethod
    def floats_for(engine) -> int:
        """Capacity that gives every trained network of `engine` a region of its own (the engine then produces its gradients
        straight into the peer buffer and the optimizers read the sums from it: no copy in, no copy out)."""
        return sum((int(g.numel()) + 63) // 64 * 64 for g in engine.grads.values())

    @classmethod
    def create(cls, max_floats: int, group=None):
        """A connected communicator, or None on EVERY rank if any rank failed (agreement over the group)."""
        comm, ok, err = None, 1, None
        multi = dist.is_initialized() and dist.get_world_size(group) > 1
        state = {"exchanged": False}
        try:
            comm = cls(max_floats, group, _state=state)
            ok = int(comm.self_test())
        except Exception as ex:          # RecnnHipError, an allocation failure, ...: this rank votes no instead of leaving its peers
            ok, err = 0, ex              # inside a collective it never joins
            if multi and not state["exchanged"]:
                # the constructor failed BEFORE the handle exchange: take part in it with a non-handle so that the other ranks'
                # constructors see the failure instead of waiting / joining garbage.  (A constructor that failed AFTER its gather --
                # connect, "a rank could not export" -- must NOT gather again: every rank runs exactly one exchange gather and one
                # vote gather)
                try:
                    every = [None] * dist.get_world_size(group)
                    dist.all_gather_object(every, 0, group=group)
                except Exception:
                    pass
        if multi:
            votes = [None] * dist.get_world_size(group)
            dist.all_gather_object(votes, ok, group=group)
            ok = min(votes)
        if not ok:
            if comm is not None:
                comm.close()
            if err is not None and not isinstance(err, L.RecnnHipError):
                raise err                # not a library failure: reported, but only after the agreement
            return None
        return comm

    def self_test(self) -> bool:
        """Three collectives on known vectors (exact in fp32), checked on this rank: a protocol that does not work on this
        machine's fabric (flags or data not visible across GPUs, a peer that never arrives: the waits are bounded) shows up here
        as a wrong sum or a reported time-out instead of inside a training run."""
        n = min(4099, self.max_floats)
        base = (torch.arange(n, dtype=torch.float32) % 7 + 1).cuda()
        want = base * (self.world * (self.world + 1) / 2)
        try:
            for rep in range(3):
                x = base * float(self.rank + 1)
                self.all_reduce(x)
                torch.cuda.synchronize()
                if not torch.equal(x, want):
                    return False
            self.check()
        except L.RecnnHipError:
            return False
        return True

    def all_reduce(self, t: torch.Tensor):
        """In-place sum over the ranks of a contiguous fp32 CUDA tensor, on the current stream (same bits on every rank)."""
        if t.dtype != torch.float32 or not t.is_cuda or not t.is_contiguous():
            raise ValueError("PeerComm.all_reduce needs a contiguous fp32 CUDA tensor")
        L.call("recnn_dp_allreduce_flat", self.handle, L.ptr(t), t.numel(), L.current_stream())
        return t

    def check(self):
        """Raises if a wait for a peer ran out since the last check (synchronises the device)."""
        L.call("recnn_comm_status", self.handle, None, None)

    def set_timeout_ms(self, ms: int):
        """Bound of every peer wait of the collectives launched / captured afterwards (default 4000 ms)."""
        L.call("recnn_comm_set_timeout_ms", self.handle, int(ms))

    def clear_status(self):
        L.call("recnn_comm_clear_status", self.handle)

    def close(self):
        if getattr(self, "handle", None) is not None:
            self.lib.recnn_comm_destroy(self.handle)
            self.handle = None


class DataParallelStepper:
    """Drives any engine exposing the phase API of recnn_amd.nn.engine.StepEngine
    (value_grads / value_apply / policy_grads / policy_apply / finish / grad_arena / value_nets / policy_every)."""

    def __init__(self, engine, rows: int, group=None, use_graphs: bool = True, always_reduce: bool = False,
                 overlap: bool = False, comm: "PeerComm | None" = None):
        if not dist.is_initialized():
            raise RuntimeError("DataParallelStepper needs an initialised torch.distributed process group "
                               "(backend 'nccl' = RCCL on ROCm)")
        self.engine = engine
        self.rows = rows
        self.group = group
        self.world = dist.get_world_size(group)
        self.scale = 1.0 / self.world
        self.always_reduce = always_reduce
        # With a PeerComm the collectives are launches INSIDE the engine's steps (csrc/comm.hip): a data-parallel run is the
        # single-GPU run graph with collective nodes -- `run` is one call per stretch, no Python between the phases of a step.
        self.comm = comm
        if comm is not None:
            if comm.world != self.world:
                raise ValueError(f"PeerComm spans {comm.world} ranks, the process group {self.world}")
            engine.set_comm(comm, self.scale)
            self.graphs = bool(use_graphs and hasattr(engine, "graph_build"))
            self.overlap = False
            self._eager_sampler = (not self.graphs) and bool(getattr(engine, "has_sampler", False))
            if self.graphs:
                engine.graph_build(rows)
            return
        # the phases between the all-reduces replay as hipGraphs when the engine offers them (StepEngine does)
        self.graphs = bool(use_graphs and hasattr(engine, "dp_graph_build"))
        # overlap (optional): the actor forward does not depend on the critic update, so it can be launched while the
        # critic gradient all-reduce (issued async on RCCL's own stream) is in flight.  Measured with one rank it costs
        # +37 us per step (the actor leaves the fused 3-network launch), so it only pays when the collective is slower
        # than that; off by default until measured on a multi-GPU node.
        self.overlap = bool(self.graphs and overlap and (self.world > 1 or always_reduce))
        if self.graphs:
            engine.dp_graph_build(rows, self.scale, self.overlap)
        # eager phases draw the step's batch from the bound sampler -- but ONLY inside this stepper's own phase calls: the
        # engine flag is raised around them (_eager_scope) and dropped again, so that a later `algo.update(batch)` on the same
        # engine evaluates the batch it was given and leaves the device cursor alone
        self._eager_sampler = (not self.graphs) and bool(getattr(engine, "has_sampler", False))

    def _allreduce(self, t: torch.Tensor):
        if self.comm is not None:
            self.comm.all_reduce(t)
        elif self.world > 1 or self.always_reduce:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)

    def step(self, t: int, learn: bool = True):
        e = self.engine
        if self.comm is not None:
            if self.graphs and learn:
                e.graph_run(t, 1)
                return
            if self._eager_sampler:
                e.sampler_eager(True)
            try:
                e.step(self.rows, learn, t)
            finally:
                if self._eager_sampler:
                    e.sampler_eager(False)
            return
        policy = learn and (t % e.policy_every == 0)
        if self.graphs and learn:
            e.dp_graph_launch(0)
            if self.overlap:
                works = [dist.all_reduce(e.grad_arena(ni), op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                         for ni in e.value_nets()]
                e.dp_graph_launch(4)                 # actor forward, concurrent with the collective
                for w in works:
                    w.wait()                         # stream-level wait, the host does not block
            else:
                for ni in e.value_nets():
                    self._allreduce(e.grad_arena(ni))
            if not policy:
                e.dp_graph_launch(1)
            else:
                e.dp_graph_launch(2)
                self._allreduce(e.grad_arena(L.NET_POLICY))
                e.dp_graph_launch(3)
            return
        if self._eager_sampler:
            e.sampler_eager(True)
        try:
            e.value_grads(self.rows, learn)
            if learn:
                for ni in e.value_nets():
                    self._allreduce(e.grad_arena(ni))
                e.value_apply(policy, self.scale)
            e.policy_grads(self.rows, policy)
            if policy:
                self._allreduce(e.grad_arena(L.NET_POLICY))
                e.policy_apply(True, self.scale)
            e.finish(self.rows, learn, policy)
        finally:
            if self._eager_sampler:
                e.sampler_eager(False)

    check_every = 16     # run() calls between two polls of the communicator's error word (a poll synchronises the device)

    def grad_sync_error(self, raise_on_all_ranks: bool = True) -> bool:
        """True when a bounded wait of the device collective ran out on ANY rank since the last poll (the sums of those steps
        are invalid and rank-divergent): every rank then raises together (a training run must not continue silently
        on corrupt gradients).  Synchronises the device and the group."""
        if self.comm is None:
            return False
        bad = 0
        try:
            self.comm.check()
        except L.RecnnHipError:
            bad = 1
        if self.world > 1:
            votes = [None] * self.world
            dist.all_gather_object(votes, bad, group=self.group)
            bad = max(votes)
        if bad and raise_on_all_ranks:
            raise L.RecnnHipError("data parallel: a gradient exchange timed out on at least one rank -- the parameters of the "
                                  "steps since the previous check are invalid (recnn_comm_status names the peers)")
        return bool(bad)

    def _poll(self):
        if self.comm is None:
            return
        self._runs_since_check = getattr(self, "_runs_since_check", 0) + 1
        if self._runs_since_check >= self.check_every:
            self._runs_since_check = 0
            self.grad_sync_error()

    def run(self, first: int, n: int, learn: bool = True):
        """`n` consecutive steps.  With phase graphs (and no overlap mode) the tail of step t and the head of step t+1
        replay as ONE graph -- one graph launch per step instead of two -- and, when the engine samples its own
        bf16 batches, the gather of step t+1 rides on step t's critic optimizer launch (two batch buffer sets)."""
        e = self.engine
        if n <= 0:
            return
        if self.comm is not None and self.graphs and learn:
            e.graph_run(first, n)
            self._poll()
            return
        if not (self.graphs and learn) or self.overlap:
            for t in range(first, first + n):
                self.step(t, learn)
            self._poll()
            return
        two = e.dp_sets() == 2
        s = 0
        e.dp_graph_launch(0)
        for i in range(n):
            t, last = first + i, i == n - 1
            for ni in e.value_nets():
                self._allreduce(e.grad_arena(ni))
            if t % e.policy_every != 0:
                e.dp_graph_launch((1 if last else 5) + 8 * s)
            else:
                e.dp_graph_launch(2 + 8 * s)
                self._allreduce(e.grad_arena(L.NET_POLICY))
                e.dp_graph_launch((3 if last else 6) + 8 * s)
            if two and not last:
                s ^= 1

    def check_replicas(self, tensors) -> float:
        """max |x - mean over ranks| over the given parameter tensors (0 when replicas agree bit for bit)."""
        worst = 0.0
        for t in tensors:
            ref = t.detach().clone()
            self._allreduce(ref)
            worst = max(worst, float((ref * self.scale - t).abs().max()))
        return worst


# ---------------------------------------------------------------------------------------------------------------------
# Item-dimension (vocabulary) parallel REINFORCE policy head (new functionality; SURVEY.md 8(f)1, BASELINE configs[4]: a
# 100k-item catalogue "on 8 x MI355X").  The reference's DiscreteActor (recnn/nn/models.py:76-99) is
#     probs = softmax(linear2(relu(linear1(state))))          linear2: [n_items, hidden]
# and its log-prob is torch.distributions.Categorical(probs).log_prob(action) = log(clamp(p_a, eps, 1 - eps))
# (models.py:103-111).  Here rank r holds rows [n0, n1) of linear2 (weights, Adam state and the [B, n_items / W] activations
# are 1 / W of the whole); linear1 is replicated.  Per forward: all-reduce(MAX) of the row maxima and all-reduce(SUM) of the
# row sums of exp (the softmax denominator), all-reduce(SUM) of the chosen action's probability (one rank owns it); per
# backward one all-reduce(SUM) of the [B, hidden] gradient that flows back into the replicated layer 1.  The catalogue-sized
# products (logits, dW2, dlogits W2) run on the HIP GEMM kernels; `ops` is swappable so the gloo tests on CPU can stand torch
# matmuls in for them.
class _HipOps:
    """The three products of the head on librecnn_hip's GEMM kernels (fp32)."""

    @staticmethod
    def linear(x, w, b, relu):
        from .nn import functional as Fh
        B, K = x.shape
        N = w.shape[0]
        Kp, ldn = Fh._r64(K), Fh._r64(N)
        xp, wp = Fh._pad(x, B, Kp), Fh._pad(w, Fh._r4(N), Kp)
        out = torch.empty(B, ldn, device=x.device)
        if ldn != N:
            out[:, N:].zero_()
        Fh._fwd(xp, Kp, wp, b.detach().float().contiguous(), out, ldn, N, relu, None)
        return out[:, :N]

    # ---- the sharded softmax's row passes (csrc/policy.hip); x = this rank's [B, ns] logits, rows 16-byte aligned, pitch % 4 == 0
    @staticmethod
    def _rows(x):
        if x.stride(1) != 1 or x.stride(0) % 4 or x.data_ptr() % 16:
            raise L.RecnnHipError("sharded softmax: logits rows must be contiguous, 16-byte aligned, pitch a multiple of 4 floats")
        return L.ptr(x), x.stride(0), x.shape[0], x.shape[1]

    @staticmethod
    def rowmax(x):
        p, ld, B, n = _HipOps._rows(x)
        m = torch.empty(B, device=x.device)
        L.call("recnn_shard_softmax_pass", p, ld, B, n, 0, L.ptr(m), None, None, L.current_stream())
        return m

    @staticmethod
    def exp_rowsum_(x, m):
        p, ld, B, n = _HipOps._rows(x)
        s = torch.empty(B, device=x.device)
        L.call("recnn_shard_softmax_pass", p, ld, B, n, 1, L.ptr(m), None, L.ptr(s), L.current_stream())
        return s

    @staticmethod
    def norm_pick_(x, ssum, local):
        p, ld, B, n = _HipOps._rows(x)
        pa = torch.empty(B, device=x.device)
        L.call("recnn_shard_softmax_pass", p, ld, B, n, 2, L.ptr(ssum), L.ptr(local), L.ptr(pa), L.current_stream())
        return pa

    @staticmethod
    def logprob_bwd(probs, local, g):
        p, ld, B, n = _HipOps._rows(probs)
        ldd = (n + 63) // 64 * 64
        buf = torch.empty(B, ldd, device=probs.device)
        if ldd != n:
            buf[:, n:].zero_()
        L.call("recnn_shard_logprob_bwd", p, ld, B, n, L.ptr(local), L.ptr(g), L.ptr(buf), ldd, L.current_stream())
        return buf[:, :n]

    @staticmethod
    def grad_w(dz, x):          # dz^T x : [N, K]
        from .nn import functional as Fh
        B, N = dz.shape
        K = x.shape[1]
        dzp, xp = Fh._pad(dz, B, Fh._r64(N)), Fh._pad(x, B, Fh._r64(K))
        out = torch.empty(N, K, device=dz.device)
        Fh._dw(dzp, N, xp, K, out)
        return out

    @staticmethod
    def grad_x(dz, w):          # dz w : [B, K], contraction over the (sharded) item axis: w^T puts it on the contiguous axis
        from .nn import functional as Fh
        B, N = dz.shape
        K = w.shape[1]
        ldn, Kp = Fh._r64(N), Fh._r64(K)

        def transposed(t):
            return Fh.transposed_rows(t, ldn)
        wt = Fh._derived_of(w, "transposed", transposed) if isinstance(w, torch.nn.Parameter) else transposed(w)
        out = torch.zeros(B, Kp, device=dz.device)
        Fh._fwd(Fh._pad(dz, B, ldn), ldn, wt, None, out, Kp, K, False, None)
        return out[:, :K]


class VocabParallelPolicyFunction(torch.autograd.Function):
    """(log_prob [B], probs_shard [B, n1 - n0]) of actions under softmax over the WHOLE catalogue; see the section header."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2s, b2s, actions, n0, group, ops):
        h = ops.linear(x, w1, b1, True)
        logits = ops.linear(h, w2s, b2s, False)
        # the softmax over the WHOLE catalogue in three passes over this rank's shard (policy.hip shard_* kernels through `ops`; round 6:
        # these were ATen amax / exp / sum / div / gather calls), the all-reduces between them are RCCL's
        m = ops.rowmax(logits)
        dist.all_reduce(m, op=dist.ReduceOp.MAX, group=group)
        ssum = ops.exp_rowsum_(logits, m)                       # logits -> exp(logits - m), in place
        dist.all_reduce(ssum, op=dist.ReduceOp.SUM, group=group)
        ns = w2s.shape[0]
        local = (actions.to(torch.int64) - n0).contiguous()
        own = (local >= 0) & (local < ns)
        pa = ops.norm_pick_(logits, ssum, local)                # ... -> this shard's probabilities, in place; p of the action if owned
        probs = logits
        dist.all_reduce(pa, op=dist.ReduceOp.SUM, group=group)
        eps = torch.finfo(torch.float32).eps
        lp = pa.clamp(eps, 1 - eps).log()
        inside = ((pa >= eps) & (pa <= 1 - eps)).float()
        ctx.save_for_backward(x, h, w1, w2s, probs, local, own, inside)
        ctx.group, ctx.ops = group, ops
        ctx.mark_non_differentiable(probs)
        return lp, probs

    @staticmethod
    def backward(ctx, dlp, _dprobs):
        x, h, w1, w2s, probs, local, own, inside = ctx.saved_tensors
        ops = ctx.ops
        g = (dlp.float() * inside).contiguous()                   # d log(clamp(p_a)) / d logit_j = [p_a inside] (delta_ja - p_j)
        dlog = ops.logprob_bwd(probs, local, g)                   # -g p (+ g at the owned action): policy.hip shard_logprob_bwd_kernel
        gw2 = ops.grad_w(dlog, h)
        gb2 = dlog.sum(0)
        dh = ops.grad_x(dlog, w2s).contiguous()
        dist.all_reduce(dh, op=dist.ReduceOp.SUM, group=ctx.group)   # the other shards' share of d loss / d h
        dz1 = dh * (h > 0)
        gw1 = ops.grad_w(dz1, x)
        gb1 = dz1.sum(0)
        gx = ops.grad_x(dz1, w1) if ctx.needs_input_grad[0] else None
        return gx, gw1, gb1, gw2, gb2, None, None, None, None


class VocabParallelDiscreteActor(torch.nn.Module):
    """DiscreteActor (recnn/nn/models.py:76-99) with linear2 sharded over the ranks of `group` along the catalogue.

    `from_full(actor)` takes this rank's rows of a replicated DiscreteActor (same init on every rank); `log_prob(state,
    actions)` scores given actions (the corrected / top-K REINFORCE path scores logged actions); `sample(state, seed)` draws
    from the full softmax by inverse CDF with the same uniforms on every rank.  `linear1`'s gradients come out replicated (no
    reduction needed), `linear2`'s are this rank's rows."""

    def __init__(self, input_dim, action_dim, hidden_size, group=None, ops=None):
        super().__init__()
        ready = dist.is_initialized()
        self.group = group
        self.world = dist.get_world_size(group) if ready else 1
        self.rank = dist.get_rank(group) if ready else 0
        per = -(-action_dim // self.world)
        self.n_items = action_dim
        self.n0, self.n1 = min(per * self.rank, action_dim), min(per * (self.rank + 1), action_dim)
        self.linear1 = torch.nn.Linear(input_dim, hidden_size)
        self.linear2 = torch.nn.Linear(hidden_size, self.n1 - self.n0)
        self.ops = ops
        self.saved_log_probs, self.rewards, self.correction, self.lambda_k = [], [], [], []
        self.action_source = {"pi": "pi", "beta": "beta"}
        self.select_action = self._select_action
        self.forced_actions = []
        self._draws = 0

    @classmethod
    def from_full(cls, actor, group=None, ops=None):
        m = cls(actor.linear1.in_features, actor.linear2.out_features, actor.linear1.out_features, group, ops)
        with torch.no_grad():
            m.linear1.weight.copy_(actor.linear1.weight)
            m.linear1.bias.copy_(actor.linear1.bias)
            m.linear2.weight.copy_(actor.linear2.weight[m.n0:m.n1])
            m.linear2.bias.copy_(actor.linear2.bias[m.n0:m.n1])
        return m.to(actor.linear1.weight.device)

    def _ops(self, x):
        if self.ops is not None:
            return self.ops
        if not x.is_cuda:
            raise L.RecnnHipError("VocabParallelDiscreteActor runs on the GPU only (no CPU fallback); tests pass ops= explicitly")
        return _HipOps

    # ---- the episode interface of DiscreteActor (recnn/nn/models.py:86-184), so that `reinforce_update` takes this module as
    # nets["policy_net"] / nets["target_policy_net"]: every "probabilities" it returns is THIS RANK'S COLUMNS [n0, n1) of the
    # softmax over the whole catalogue -- what VocabParallelCritic contracts with its own columns of layer 1.
    def forward(self, inputs):
        with torch.no_grad():
            _, probs = self.log_prob(inputs, torch.zeros(inputs.shape[0], dtype=torch.int64, device=inputs.device))
        return probs

    def gc(self):
        del self.rewards[:]
        del self.saved_log_probs[:]
        del self.correction[:]
        del self.lambda_k[:]

    def _draw(self, state):
        """the policy's own action: queued in `forced_actions` (replays, parity tests) or sampled, the same on every rank"""
        if self.forced_actions:
            return self.forced_actions.pop(0).to(state.device)
        self._draws += 1
        return self.sample(state, seed=(torch.initial_seed() + 7919 * self._draws) % (1 << 62))

    def _select_action(self, state, **kwargs):
        log_prob, probs = self.log_prob(state, self._draw(state))
        self.saved_log_probs.append(log_prob)
        return probs

    def pi_beta_sample(self, state, beta, action, **kwargs):
        """models.py:113-141 with a REPLICATED behaviour policy: `beta(state, action=...)` returns probabilities over the whole
        catalogue on every rank (the notebook's Beta is one Linear: it is not sharded here); its action is drawn by inverse CDF
        from a generator seeded alike on every rank."""
        beta_probs = beta(state.detach(), action=action).detach()
        self._draws += 1
        u = torch.rand(state.shape[0], generator=torch.Generator().manual_seed((torch.initial_seed() + 7919 * self._draws) % (1 << 62)))
        cdf = beta_probs.cumsum(1)
        beta_action = torch.searchsorted((cdf / cdf[:, -1:]).contiguous(), u.to(state.device)[:, None].contiguous(), right=True)[:, 0]
        beta_action = beta_action.clamp(max=beta_probs.shape[1] - 1)
        eps = torch.finfo(torch.float32).eps

        def beta_lp(act):
            pr = beta_probs / beta_probs.sum(-1, keepdim=True)
            return torch.log(pr.clamp(eps, 1 - eps)).gather(-1, act.reshape(-1, 1).to(torch.int64)).reshape(act.shape)
        pi_action = beta_action if self.action_source["pi"] == "beta" else self._draw(state)
        pi_log_prob, pi_probs = self.log_prob(state, pi_action)
        beta_log_prob = beta_lp(beta_action if self.action_source["beta"] == "beta" else pi_action)
        return pi_log_prob, beta_log_prob, pi_probs

    def _select_action_with_correction(self, state, beta, action, writer, step, **kwargs):
        pi_log_prob, beta_log_prob, pi_probs = self.pi_beta_sample(state, beta, action)
        corr = torch.exp(pi_log_prob) / torch.exp(beta_log_prob)
        writer.add_histogram("correction", corr, step)
        writer.add_histogram("pi_log_prob", pi_log_prob, step)
        writer.add_histogram("beta_log_prob", beta_log_prob, step)
        self.correction.append(corr)
        self.saved_log_probs.append(pi_log_prob)
        return pi_probs

    def _select_action_with_TopK_correction(self, state, beta, action, K, writer, step, **kwargs):
        pi_log_prob, beta_log_prob, pi_probs = self.pi_beta_sample(state, beta, action)
        corr = torch.exp(pi_log_prob) / torch.exp(beta_log_prob)
        l_k = K * (1 - torch.exp(pi_log_prob)) ** (K - 1)
        writer.add_histogram("correction", corr, step)
        writer.add_histogram("l_k", l_k, step)
        writer.add_histogram("pi_log_prob", pi_log_prob, step)
        writer.add_histogram("beta_log_prob", beta_log_prob, step)
        self.correction.append(corr)
        self.lambda_k.append(l_k)
        self.saved_log_probs.append(pi_log_prob)
        return pi_probs

    def log_prob(self, state, actions):
        """(log_prob [B], this rank's columns [n0, n1) of the probabilities)."""
        return VocabParallelPolicyFunction.apply(state.float(), self.linear1.weight, self.linear1.bias, self.linear2.weight,
                                                 self.linear2.bias, actions, self.n0, self.group, self._ops(state))

    @torch.no_grad()
    def sample(self, state, seed: int):
        """Actions ~ softmax over the whole catalogue: u ~ U[0, 1) per row (the same on every rank: generator seeded with
        `seed`), the rank whose cumulative mass interval contains u finds the item inside its shard."""
        B = state.shape[0]
        _, probs = self.log_prob(state, torch.zeros(B, dtype=torch.int64, device=state.device))
        u = torch.rand(B, generator=torch.Generator().manual_seed(int(seed))).to(state.device)
        mass = torch.zeros(B, self.world, device=state.device)
        mass[:, self.rank] = probs.sum(1)
        dist.all_reduce(mass, op=dist.ReduceOp.SUM, group=self.group)
        upper = mass.cumsum(1)
        lower = upper - mass
        lo, hi = lower[:, self.rank], upper[:, self.rank]
        last = self.rank == self.world - 1
        mine = (u >= lo) & ((u < hi) | last)                     # (rounding may leave u above the last upper bound)
        idx = torch.searchsorted(probs.cumsum(1).contiguous(), (u - lo)[:, None].contiguous(), right=True)[:, 0]
        idx = idx.clamp(max=probs.shape[1] - 1) + self.n0
        out = torch.where(mine, idx + 1, torch.zeros_like(idx))
        dist.all_reduce(out, op=dist.ReduceOp.SUM, group=self.group)
        return out - 1


# ---------------------------------------------------------------------------------------------------------------------------
# The REINFORCE critic over [state | action distribution] (recnn/nn/models.py:187-213 with action_dim = n_items): its first layer
# is [hidden, 1290 + n_items] -- as wide as the actor's head.  Sharded the same way: rank r keeps the action columns [n0, n1) of
# linear1 (with their gradient and optimizer state) and contracts them with ITS columns of the action distribution (what
# VocabParallelDiscreteActor returns) or of a one-hot batch action; the [B, hidden] partial products are summed over the ranks
# (one all-reduce of B x hidden floats per forward) and enter the replicated rest of the MLP before the first relu.
class _ShardMatmul(torch.autograd.Function):
    """a_shard [B, ns] x w_shard^T [ns, H] -> this rank's partial of the layer-1 pre-activation (local backward)"""

    @staticmethod
    def forward(ctx, a, w, ops):
        ctx.save_for_backward(a, w)
        ctx.ops = ops
        return ops.linear(a, w, torch.zeros(w.shape[0], device=w.device), False).contiguous()

    @staticmethod
    def backward(ctx, dz):
        a, w = ctx.saved_tensors
        dz = dz.contiguous()
        ga = ctx.ops.grad_x(dz, w) if ctx.needs_input_grad[0] else None
        gw = ctx.ops.grad_w(dz, a) if ctx.needs_input_grad[1] else None
        return ga, gw, None


class _AllReduceSum(torch.autograd.Function):
    """y = sum over ranks of the partials; dL/d partial_r = dL/dy, which every rank computes alike from replicated layers"""

    @staticmethod
    def forward(ctx, partial, group):
        out = partial.clone()
        dist.all_reduce(out, op=dist.ReduceOp.SUM, group=group)
        return out

    @staticmethod
    def backward(ctx, g):
        return g, None


class VocabParallelCritic(torch.nn.Module):
    """Critic(input_dim, n_items, hidden) with linear1's ACTION columns sharded over the ranks of `group`.

    Parameters: `linear1_state` (weight [H, S] and the layer's bias; replicated), `w1_action` [H, n1 - n0] (this rank's columns),
    `linear2`, `linear3` (replicated).  Replicated layers see the same inputs on every rank, so their gradients come out
    identical without a reduction.  `forward(state, action)`: `action` is [B, n_items] (sliced here; one-hot rows made by
    `functional.onehot_rows` are gathered instead of contracted) or already this rank's [B, n1 - n0] columns."""

    def __init__(self, input_dim, action_dim, hidden_size, group=None, ops=None):
        super().__init__()
        ready = dist.is_initialized()
        self.group = group
        self.world = dist.get_world_size(group) if ready else 1
        self.rank = dist.get_rank(group) if ready else 0
        per = -(-action_dim // self.world)
        self.n_items = action_dim
        self.n0, self.n1 = min(per * self.rank, action_dim), min(per * (self.rank + 1), action_dim)
        self.drop_layer = torch.nn.Dropout(p=0.5)
        self.linear1_state = torch.nn.Linear(input_dim, hidden_size)
        self.w1_action = torch.nn.Parameter(torch.zeros(hidden_size, self.n1 - self.n0))
        self.linear2 = torch.nn.Linear(hidden_size, hidden_size)
        self.linear3 = torch.nn.Linear(hidden_size, 1)
        self.ops = ops

    @classmethod
    def from_full(cls, critic, state_dim, group=None, ops=None):
        """this rank's share of a replicated Critic(state_dim, n_items, hidden) (same init on every rank)"""
        n_items = critic.linear1.weight.shape[1] - state_dim
        m = cls(state_dim, n_items, critic.linear1.weight.shape[0], group, ops)
        with torch.no_grad():
            m.linear1_state.weight.copy_(critic.linear1.weight[:, :state_dim])
            m.linear1_state.bias.copy_(critic.linear1.bias)
            m.w1_action.copy_(critic.linear1.weight[:, state_dim + m.n0:state_dim + m.n1])
            for a, b in ((m.linear2, critic.linear2), (m.linear3, critic.linear3)):
                a.weight.copy_(b.weight)
                a.bias.copy_(b.bias)
        return m.to(critic.linear1.weight.device)

    def _action_part(self, action, ops):
        from .nn import functional as Fh
        ns = self.n1 - self.n0
        idx = Fh.onehot_index_of(action)
        if idx is not None:          # a one-hot batch action: the owned rows' weight columns, zeros elsewhere
            local = idx.to(torch.int64) - self.n0
            own = (local >= 0) & (local < ns)
            cols = self.w1_action.index_select(1, local.clamp(0, max(ns - 1, 0))).t()
            part = torch.where(own[:, None], cols, torch.zeros_like(cols))
        else:
            a = action[:, self.n0:self.n1] if action.shape[1] == self.n_items and self.world > 1 else action
            if a.shape[1] != ns:
                raise ValueError(f"VocabParallelCritic: action has {action.shape[1]} columns, expected {self.n_items} or this rank's {ns}")
            part = _ShardMatmul.apply(a.float().contiguous(), self.w1_action, ops)
        return _AllReduceSum.apply(part, self.group)

    def forward(self, state, action):
        ops = self.ops
        if ops is None:
            if not state.is_cuda:
                raise L.RecnnHipError("VocabParallelCritic runs on the GPU only (no CPU fallback); tests pass ops= explicitly")
            ops = _HipOps
        part = self._action_part(action, ops)
        if ops is _HipOps:
            from .nn import functional as Fh
            return Fh.MLPFunction.apply(state.float(), self.linear1_state.weight, self.linear1_state.bias, self.linear2.weight,
                                        self.linear2.bias, self.linear3.weight, self.linear3.bias, self.training, torch.initial_seed(),
                                        Fh._take_forced_masks(self, self.training), part)
        h = self.drop_layer(torch.relu(self.linear1_state(state.float()) + part))
        h = self.drop_layer(torch.relu(self.linear2(h)))
        return self.linear3(h)
