"""Worker of tests/test_gpu_reinforce.py::test_vocab_parallel_head_on_hip_gemms: 2 ranks (gloo; both on cuda:0), the
item-dimension-parallel REINFORCE head on the HIP GEMM kernels at a 100k-item catalogue against the unsharded HIP
DiscreteActor (recnn_amd.nn.DiscreteActor = recnn/nn/models.py:76-111) on the same weights, states and actions."""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out_dir = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo")
    import recnn_amd
    from recnn_amd.parallel import VocabParallelDiscreteActor
    S, N, H, B = 1290, 100_000, 256, 64
    torch.manual_seed(0)                       # same replicated init on every rank
    full = recnn_amd.nn.DiscreteActor(S, N, H).to(dev)
    gen = torch.Generator().manual_seed(1)
    x = torch.randn(B, S, generator=gen).to(dev)
    act = torch.randint(0, N, (B,), generator=gen).to(dev)
    coef = torch.randn(B, generator=gen).to(dev)
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    shard = VocabParallelDiscreteActor.from_full(full)
    xs = x.clone().requires_grad_(True)
    lp, probs = shard.log_prob(xs, act)
    (lp * coef).sum().backward()
    torch.cuda.synchronize()
    shard_peak = torch.cuda.max_memory_allocated() - base
    sampled = shard.sample(x, seed=5)
    # the unsharded HIP head
    from recnn_amd.nn import functional as F_hip
    xf = x.clone().requires_grad_(True)
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    probs_f, _, lp_f = F_hip.discrete_policy(xf, full, actions=act)
    (lp_f * coef).sum().backward()
    torch.cuda.synchronize()
    full_peak = torch.cuda.max_memory_allocated() - base

    def rel(a, b):
        return float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30))
    n0, n1 = shard.n0, shard.n1
    u = torch.rand(B, generator=torch.Generator().manual_seed(5)).to(dev)
    want = torch.searchsorted(probs_f.detach().cumsum(1).contiguous(), u[:, None].contiguous(), right=True)[:, 0].clamp(max=N - 1)
    res = {"rank": rank, "n0": n0, "n1": n1, "lp": rel(lp, lp_f), "probs": rel(probs, probs_f[:, n0:n1]),
           "gw1": rel(shard.linear1.weight.grad, full.linear1.weight.grad), "gb1": rel(shard.linear1.bias.grad, full.linear1.bias.grad),
           "gw2": rel(shard.linear2.weight.grad, full.linear2.weight.grad[n0:n1]),
           "gb2": rel(shard.linear2.bias.grad, full.linear2.bias.grad[n0:n1]), "gx": rel(xs.grad, xf.grad),
           "sample_mismatch": int((sampled != want).sum()), "shard_peak_mb": shard_peak / 2**20, "full_peak_mb": full_peak / 2**20}
    every = [None] * world
    dist.all_gather_object(every, res)
    if rank == 0:
        os.makedirs(out_dir, exist_ok=True)
        json.dump(every, open(os.path.join(out_dir, "vp2.json"), "w"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
