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

import torch
import torch.distributed as dist

from . import _lib as L

__all__ = ["DataParallelStepper", "PeerComm", "shard_users"]


def shard_users(perm: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    """Rank r's share of an epoch permutation of user slots."""
    return perm[rank::world].contiguous()


class PeerComm:
    """The device-side gradient exchange `recnn_dp_allreduce_flat` (csrc/comm.hip): one peer-mapped buffer per rank (hipIpc;
    one process per GPU of ONE node, world <= 8), a two-shot all-reduce as a single kernel launch -- a graph node.

    The handles travel over the given torch.distributed group once, at construction (`all_gather_object`); afterwards the
    group is not used again.  `PeerComm.create` returns None on every rank when any rank cannot map its peers (another node,
    IPC unavailable): the caller then stays on RCCL (`dist.all_reduce`)."""

    def __init__(self, max_floats: int, group=None):
        import ctypes as C
        self.lib = L.load()
        self.group = group
        ready = dist.is_initialized()
        self.world = dist.get_world_size(group) if ready else 1
        self.rank = dist.get_rank(group) if ready else 0
        h = C.c_void_p()
        L.call("recnn_comm_create", self.world, self.rank, int(max_floats), C.byref(h))
        self.handle = h
        self.max_floats = int(max_floats)
        if self.world > 1:
            nb = int(self.lib.recnn_comm_handle_bytes())
            mine = C.create_string_buffer(nb)
            L.call("recnn_comm_export", self.handle, mine, nb)
            every = [None] * self.world
            dist.all_gather_object(every, bytes(mine.raw), group=group)
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
        comm, ok = None, 1
        try:
            comm = cls(max_floats, group)
        except L.RecnnHipError:
            ok = 0
        if dist.is_initialized() and dist.get_world_size(group) > 1:
            votes = [None] * dist.get_world_size(group)
            dist.all_gather_object(votes, ok, group=group)
            ok = min(votes)
        if not ok:
            if comm is not None:
                comm.close()
            return None
        return comm

    def all_reduce(self, t: torch.Tensor):
        """In-place sum over the ranks of a contiguous fp32 CUDA tensor, on the current stream (same bits on every rank)."""
        if t.dtype != torch.float32 or not t.is_cuda or not t.is_contiguous():
            raise ValueError("PeerComm.all_reduce needs a contiguous fp32 CUDA tensor")
        L.call("recnn_dp_allreduce_flat", self.handle, L.ptr(t), t.numel(), L.current_stream())
        return t

    def check(self):
        """Raises if a wait for a peer ran out since the last check (synchronises the device)."""
        L.call("recnn_comm_status", self.handle, None, None)

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

    def run(self, first: int, n: int, learn: bool = True):
        """`n` consecutive steps.  With phase graphs (and no overlap mode) the tail of step t and the head of step t+1
        replay as ONE graph -- one graph launch per step instead of two -- and, when the engine samples its own
        bf16 batches, the gather of step t+1 rides on step t's critic optimizer launch (two batch buffer sets)."""
        e = self.engine
        if n <= 0:
            return
        if self.comm is not None and self.graphs and learn:
            e.graph_run(first, n)
            return
        if not (self.graphs and learn) or self.overlap:
            for t in range(first, first + n):
                self.step(t, learn)
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
