"""World-size-2 / 3 gloo tests (CPU) of the item-dimension-parallel REINFORCE head (recnn_amd/parallel.py
VocabParallelDiscreteActor) with torch matmuls standing in for the HIP GEMMs: log-probs, probabilities, every gradient and the
inverse-CDF sampler equal the unsharded DiscreteActor arithmetic of the reference (recnn/nn/models.py:76-111: softmax head,
Categorical(probs).log_prob)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

S, H, N, B = 19, 24, 103, 9      # 103 items: uneven shards (52 + 51; 35 + 35 + 33)


def _full(seed=0):
    gen = torch.Generator().manual_seed(seed)
    w1, b1 = torch.randn(H, S, generator=gen) * 0.3, torch.randn(H, generator=gen) * 0.1
    w2, b2 = torch.randn(N, H, generator=gen) * 0.5, torch.randn(N, generator=gen) * 0.1
    x = torch.randn(B, S, generator=gen)
    act = torch.randint(0, N, (B,), generator=gen)
    coef = torch.randn(B, generator=gen)
    return w1, b1, w2, b2, x, act, coef


def _reference():
    w1, b1, w2, b2, x, act, coef = _full()
    ps = [p.clone().requires_grad_(True) for p in (w1, b1, w2, b2)]
    xg = x.clone().requires_grad_(True)
    probs = torch.softmax(torch.relu(xg @ ps[0].t() + ps[1]) @ ps[2].t() + ps[3], dim=1)
    lp = torch.distributions.Categorical(probs).log_prob(act)
    (lp * coef).sum().backward()
    u = torch.rand(B, generator=torch.Generator().manual_seed(77))
    sampled = torch.searchsorted(probs.detach().cumsum(1).contiguous(), u[:, None].contiguous(), right=True)[:, 0].clamp(max=N - 1)
    return lp.detach(), probs.detach(), [p.grad for p in ps], xg.grad, sampled


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from recnn_amd.parallel import VocabParallelDiscreteActor
    from tests.helpers import TorchOps as _TorchOps
    w1, b1, w2, b2, x, act, coef = _full()
    m = VocabParallelDiscreteActor(S, N, H, ops=_TorchOps)
    with torch.no_grad():
        m.linear1.weight.copy_(w1); m.linear1.bias.copy_(b1)
        m.linear2.weight.copy_(w2[m.n0:m.n1]); m.linear2.bias.copy_(b2[m.n0:m.n1])
    xg = x.clone().requires_grad_(True)
    lp, probs = m.log_prob(xg, act)
    (lp * coef).sum().backward()
    sampled = m.sample(x, seed=77)
    q.put((rank, m.n0, m.n1, lp.detach().numpy().copy(), probs.numpy().copy(), m.linear1.weight.grad.numpy().copy(),
           m.linear1.bias.grad.numpy().copy(), m.linear2.weight.grad.numpy().copy(), m.linear2.bias.grad.numpy().copy(),
           xg.grad.numpy().copy(), sampled.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_vocab_parallel_head_equals_the_unsharded_head(world):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = sorted((q.get(timeout=120) for _ in range(world)), key=lambda o: o[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    lp, probs, (gw1, gb1, gw2, gb2), gx, sampled = _reference()
    assert outs[0][1] == 0 and outs[-1][2] == N and all(a[2] == b[1] for a, b in zip(outs, outs[1:]))   # shards tile the catalogue
    for rank, n0, n1, lp_r, probs_r, gw1_r, gb1_r, gw2_r, gb2_r, gx_r, sampled_r in outs:
        np.testing.assert_allclose(lp_r, lp.numpy(), rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(probs_r, probs[:, n0:n1].numpy(), rtol=1e-5, atol=1e-8)
        np.testing.assert_allclose(gw1_r, gw1.numpy(), rtol=1e-4, atol=1e-6)          # replicated layer: same on every rank
        np.testing.assert_allclose(gb1_r, gb1.numpy(), rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(gw2_r, gw2[n0:n1].numpy(), rtol=1e-4, atol=1e-6)   # this rank's rows
        np.testing.assert_allclose(gb2_r, gb2[n0:n1].numpy(), rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(gx_r, gx.numpy(), rtol=1e-4, atol=1e-6)
        assert np.array_equal(sampled_r, sampled.numpy())                             # every rank returns the same draws


# ---------------------------------------------------------------------------------------------------------------------------
# reinforce_update with BOTH catalogue-wide layers sharded: VocabParallelDiscreteActor as policy / target
# policy, VocabParallelCritic as critic / target critic -- the reference's function body, unchanged, on every rank.
RS, RH, RN, RB, RSTEPS, RPS = 11, 16, 37, 6, 12, 5       # 37 items: shards of 19 + 18


def _reinforce_setup(seed=4):
    gen = torch.Generator().manual_seed(seed)
    pol = {"w1": torch.randn(RH, RS, generator=gen) * 0.3, "b1": torch.randn(RH, generator=gen) * 0.1,
           "w2": torch.randn(RN, RH, generator=gen) * 0.5, "b2": torch.randn(RN, generator=gen) * 0.1}
    val = {"w1": torch.randn(RH, RS + RN, generator=gen) * 0.3, "b1": torch.randn(RH, generator=gen) * 0.1,
           "w2": torch.randn(RH, RH, generator=gen) * 0.3, "b2": torch.randn(RH, generator=gen) * 0.1,
           "w3": torch.randn(1, RH, generator=gen) * 0.3, "b3": torch.randn(1, generator=gen) * 0.1}
    batches = []
    for _ in range(2):
        idx = torch.randint(0, RN, (RB,), generator=gen)
        batches.append({"state": torch.randn(RB, RS, generator=gen), "action_idx": idx, "reward": torch.randn(RB, generator=gen),
                        "next_state": torch.randn(RB, RS, generator=gen), "done": (torch.rand(RB, generator=gen) < 0.2).float()})
    draws = torch.randint(0, RN, (RSTEPS, RB), generator=gen)
    return pol, val, batches, draws


def _reinforce_reference():
    from oracle import reinforce_oracle as R
    pol, val, batches, draws = _reinforce_setup()
    st = R.ReinforceState.create({k: v.clone() for k, v in pol.items()}, {k: v.clone() for k, v in val.items()},
                                 R.AdamDict(R.POLICY_ORDER, lr=1e-2), R.AdamDict(("w1", "b1", "w2", "b2", "w3", "b3"), lr=1e-2),
                                 method="basic", policy_step=RPS, soft_tau=0.05)
    losses = []
    for t in range(RSTEPS):
        b = batches[t % 2]
        onehot = torch.zeros(RB, RN)
        onehot[torch.arange(RB), b["action_idx"]] = 1
        out = R.reinforce_step(st, dict(b, action=onehot), draws[t], [None] * 4, t)
        losses.append((out["value"], out["policy"]))
    return st, losses


def _reinforce_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    import recnn_amd
    from recnn_amd.nn.update import reinforce as RU
    from recnn_amd.nn.update.reinforce import ChooseREINFORCE, reinforce_update
    from recnn_amd.parallel import VocabParallelCritic, VocabParallelDiscreteActor
    from tests.helpers import TorchOps as _TorchOps

    def torch_soft_update(net, target, soft_tau=1e-2):       # the HIP soft-update kernel is GPU-only; this test checks the sharding
        with torch.no_grad():
            for tp, p in zip(target.parameters(), net.parameters()):
                tp.copy_(tp * (1.0 - soft_tau) + p * soft_tau)
    RU.utils.soft_update = torch_soft_update
    pol, val, batches, draws = _reinforce_setup()

    def actor():
        m = VocabParallelDiscreteActor(RS, RN, RH, ops=_TorchOps)
        with torch.no_grad():
            m.linear1.weight.copy_(pol["w1"]); m.linear1.bias.copy_(pol["b1"])
            m.linear2.weight.copy_(pol["w2"][m.n0:m.n1]); m.linear2.bias.copy_(pol["b2"][m.n0:m.n1])
        return m

    def critic():
        m = VocabParallelCritic(RS, RN, RH, ops=_TorchOps)
        with torch.no_grad():
            m.linear1_state.weight.copy_(val["w1"][:, :RS]); m.linear1_state.bias.copy_(val["b1"])
            m.w1_action.copy_(val["w1"][:, RS + m.n0:RS + m.n1])
            m.linear2.weight.copy_(val["w2"]); m.linear2.bias.copy_(val["b2"])
            m.linear3.weight.copy_(val["w3"]); m.linear3.bias.copy_(val["b3"])
        return m.eval()                                      # (no dropout: the oracle run above passes no masks)
    nets = {"policy_net": actor(), "target_policy_net": actor(), "value_net": critic(), "target_value_net": critic()}
    opt = {"policy_optimizer": torch.optim.Adam(nets["policy_net"].parameters(), lr=1e-2),
           "value_optimizer": torch.optim.Adam(nets["value_net"].parameters(), lr=1e-2)}
    params = {"reinforce": ChooseREINFORCE(ChooseREINFORCE.basic_reinforce), "K": 10, "gamma": 0.99, "min_value": -10, "max_value": 10,
              "policy_step": RPS, "soft_tau": 0.05}
    n0, n1 = nets["policy_net"].n0, nets["policy_net"].n1
    losses = []
    for t in range(RSTEPS):
        b = batches[t % 2]
        onehot = torch.zeros(RB, RN)
        onehot[torch.arange(RB), b["action_idx"]] = 1
        if t % 3 == 0:
            onehot = onehot[:, n0:n1].contiguous()           # a batch action may also arrive as this rank's columns
        nets["policy_net"].forced_actions[:] = [draws[t]]
        out = reinforce_update({"state": b["state"], "action": onehot, "reward": b["reward"], "next_state": b["next_state"],
                                "done": b["done"]}, params, nets, opt, device=torch.device("cpu"), step=t)
        losses.append(None if out is None else (out["value"], out["policy"]))
    snap = {k: {n: p.detach().numpy().copy() for n, p in nets[k].named_parameters()} for k in nets}
    q.put((rank, n0, n1, losses, snap))
    dist.barrier()
    dist.destroy_process_group()


def test_reinforce_update_with_sharded_actor_and_critic_equals_the_unsharded_oracle():
    world = 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_reinforce_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = sorted((q.get(timeout=180) for _ in range(world)), key=lambda o: o[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    st, ref_losses = _reinforce_reference()
    updates = [t for t in range(RSTEPS) if t % RPS == 0 and t > 0]
    assert updates == [5, 10]
    for rank, n0, n1, losses, snap in outs:
        for t in range(RSTEPS):
            if t in updates:
                np.testing.assert_allclose(losses[t], ref_losses[t], rtol=2e-4, atol=1e-6)
            else:
                assert losses[t] is None
        for net, ref in (("policy_net", st.policy), ("target_policy_net", st.target_policy)):
            np.testing.assert_allclose(snap[net]["linear1.weight"], ref["w1"].numpy(), rtol=2e-4, atol=2e-6)
            np.testing.assert_allclose(snap[net]["linear1.bias"], ref["b1"].numpy(), rtol=2e-4, atol=2e-6)
            np.testing.assert_allclose(snap[net]["linear2.weight"], ref["w2"][n0:n1].numpy(), rtol=2e-4, atol=2e-6)
            np.testing.assert_allclose(snap[net]["linear2.bias"], ref["b2"][n0:n1].numpy(), rtol=2e-4, atol=2e-6)
        for net, ref in (("value_net", st.value), ("target_value_net", st.target_value)):
            np.testing.assert_allclose(snap[net]["linear1_state.weight"], ref["w1"][:, :RS].numpy(), rtol=2e-4, atol=2e-6)
            np.testing.assert_allclose(snap[net]["w1_action"], ref["w1"][:, RS + n0:RS + n1].numpy(), rtol=2e-4, atol=2e-6)
            np.testing.assert_allclose(snap[net]["linear1_state.bias"], ref["b1"].numpy(), rtol=2e-4, atol=2e-6)
            for k, name in (("w2", "linear2.weight"), ("b2", "linear2.bias"), ("w3", "linear3.weight"), ("b3", "linear3.bias")):
                np.testing.assert_allclose(snap[net][name], ref[k].numpy(), rtol=2e-4, atol=2e-6)
    # replicated layers hold the same bits on both ranks (same inputs, same arithmetic: no broadcast needed)
    for net in ("policy_net", "value_net"):
        for name in ("linear1.weight",) if net == "policy_net" else ("linear1_state.weight", "linear2.weight", "linear3.weight"):
            assert np.array_equal(outs[0][4][net][name], outs[1][4][net][name]), (net, name)
