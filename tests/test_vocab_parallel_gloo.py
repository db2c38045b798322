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
    from recnn_amd.parallel import VocabParallelDiscreteActor, _TorchOps
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
