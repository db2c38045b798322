"""World-size-2 gloo test (CPU) of the data-parallel orchestration in recnn_amd/parallel.py.

The HIP engine cannot run here, so a CPU stand-in built on the oracle implements the same phase API
(value_grads / value_apply / policy_grads / policy_apply / finish).  What is proven: N ranks x B/N rows,
all-reducing the flat gradient arenas where DataParallelStepper does, equal 1 rank x B rows; the L1 clip
quirk is applied to the REDUCED gradient; replicas stay identical without a broadcast.
"""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import recnn_oracle as O

S, A, H, B = 27, 8, 16, 12


def _mk(gen, inp, out):
    return {"w1": torch.randn(H, inp, generator=gen) * 0.2, "b1": torch.randn(H, generator=gen) * 0.1,
            "w2": torch.randn(H, H, generator=gen) * 0.2, "b2": torch.randn(H, generator=gen) * 0.1,
            "w3": torch.randn(out, H, generator=gen) * 0.3, "b3": torch.randn(out, generator=gen) * 0.1}


def _problem():
    gen = torch.Generator().manual_seed(0)
    actor, critic = _mk(gen, S, A), _mk(gen, S + A, 1)
    steps = []
    for _ in range(4):
        batch = {"state": torch.randn(B, S, generator=gen), "action": torch.randn(B, A, generator=gen),
                 "reward": torch.randn(B, generator=gen), "next_state": torch.randn(B, S, generator=gen),
                 "done": (torch.rand(B, generator=gen) < 0.2).float()}
        masks = [(torch.rand(B, H, generator=gen) < 0.5).to(torch.uint8) for _ in range(6)]
        steps.append((batch, masks))
    return actor, critic, steps


class OracleEngine:
    """Phase API of StepEngine on the CPU oracle (flat gradient arenas in the canonical [w1|b1|..|b3] layout)."""

    def __init__(self, actor, critic, lr=1e-2):
        self.st = O.DDPGState.create(O.clone_params(actor), O.clone_params(critic), O.AdamState(lr=lr), O.AdamState(lr=lr))
        self.st.params["policy_step"] = 2
        self.policy_every = 2
        self.flat = {0: torch.zeros(sum(v.numel() for v in actor.values())),
                     2: torch.zeros(sum(v.numel() for v in critic.values()))}
        self.losses = {}

    def value_nets(self):
        return (2,)

    def grad_arena(self, ni):
        return self.flat[ni]

    def load(self, batch, masks):
        self.batch, self.masks = batch, masks

    def _pack(self, ni, g):
        torch.cat([g[k].reshape(-1) for k in O.PARAM_ORDER], out=self.flat[ni])

    def _unpack(self, ni, like):
        out, off = {}, 0
        for k in O.PARAM_ORDER:
            n = like[k].numel()
            out[k] = self.flat[ni][off:off + n].view_as(like[k]).clone()
            off += n
        return out

    def value_grads(self, rows, learn=True):
        st, b, m = self.st, self.batch, self.masks
        s, a = b["state"], b["action"]
        r, d = b["reward"].reshape(-1, 1), b["done"].reshape(-1, 1)
        na, _ = O.actor_forward(st.target_policy, b["next_state"])
        tq, _ = O.critic_forward(st.target_value, b["next_state"], na)
        y = torch.clamp(O.temporal_difference(r, d, st.params["gamma"], tq), st.params["min_value"], st.params["max_value"])
        q, cache = O.critic_forward(st.value, s, a, m[0], m[1])
        self.losses["value"] = float(((q - y) ** 2).mean())
        g, _, _ = O.mlp_backward(st.value, cache, (q - y) * (2.0 / rows))
        self._pack(2, g)

    def value_apply(self, soft, grad_scale):
        st = self.st
        O.adam_step(st.value, self._unpack(2, st.value), st.value_opt, grad_scale=grad_scale)
        if soft:
            O.soft_update(st.value, st.target_value, st.params["soft_tau"])

    def policy_grads(self, rows, backward):
        st, b, m = self.st, self.batch, self.masks
        ga, pc = O.actor_forward(st.policy, b["state"], m[2], m[3])
        q, qc = O.critic_forward(st.value, b["state"], ga, m[4], m[5])
        self.losses["policy"] = float(-q.mean())
        if backward:
            _, dxa, _ = O.mlp_backward(st.value, qc, torch.full_like(q, -1.0 / rows), need_dx=True, need_dw=False)
            g, _, _ = O.mlp_backward(st.policy, pc, dxa[:, S:])
            self._pack(0, g)

    def policy_apply(self, soft, grad_scale):
        st = self.st
        g = {k: v * grad_scale for k, v in self._unpack(0, st.policy).items()}
        coef = O.clip_grad_quirk_scale(g)                       # on the REDUCED, averaged gradient
        O.adam_step(st.policy, g, st.policy_opt, grad_scale=coef)
        if soft:
            O.soft_update(st.policy, st.target_policy, st.params["soft_tau"])

    def finish(self, rows, value_stepped, policy_stepped):
        pass


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from recnn_amd.parallel import DataParallelStepper, shard_users
    actor, critic, steps = _problem()
    eng = OracleEngine(actor, critic)
    dp = DataParallelStepper(eng, B // world)
    lo, hi = rank * (B // world), (rank + 1) * (B // world)
    for t, (batch, masks) in enumerate(steps):
        eng.load({k: v[lo:hi] for k, v in batch.items()}, [m[lo:hi] for m in masks])
        dp.step(t)
    drift = dp.check_replicas(list(eng.st.policy.values()) + list(eng.st.value.values()) + list(eng.st.target_value.values()))
    assert shard_users(torch.arange(10), rank, world).tolist() == list(range(rank, 10, world))
    if rank == 0:
        # numpy, not tensors: a tensor travels as a shared fd that dies with this process if the parent is slow
        q.put(({k: v.numpy().copy() for k, v in eng.st.policy.items()}, {k: v.numpy().copy() for k, v in eng.st.value.items()},
               {k: v.numpy().copy() for k, v in eng.st.target_policy.items()}, float(drift)))
    dist.destroy_process_group()


def test_two_ranks_equal_one_rank_on_the_full_batch():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    pol, val, tpol, drift = q.get(timeout=120)
    pol, val, tpol = ({k: torch.from_numpy(v) for k, v in d.items()} for d in (pol, val, tpol))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert drift == 0.0                                           # replicas bit-identical, no broadcast needed
    # single process, full batch, through the plain oracle step
    actor, critic, steps = _problem()
    st = O.DDPGState.create(O.clone_params(actor), O.clone_params(critic), O.AdamState(lr=1e-2), O.AdamState(lr=1e-2))
    st.params["policy_step"] = 2
    for t, (batch, masks) in enumerate(steps):
        O.ddpg_step(st, batch, masks, step=t)
    for k in O.PARAM_ORDER:
        assert torch.allclose(pol[k], st.policy[k], rtol=2e-4, atol=2e-6), k
        assert torch.allclose(val[k], st.value[k], rtol=2e-4, atol=2e-6), k
        assert torch.allclose(tpol[k], st.target_policy[k], rtol=2e-4, atol=2e-6), k
