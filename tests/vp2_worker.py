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


def run_update(out_dir, rank, world, dev):
    """`reinforce_update` with the actor's head AND the critic's first layer sharded over 2 ranks (HIP GEMMs, gloo all-reduces of
    [B, hidden] partials) against the same function on the unsharded recnn_amd.nn.DiscreteActor / Critic in this process."""
    import copy
    import recnn_amd
    from recnn_amd.nn import functional as F_hip
    from recnn_amd.nn.update.reinforce import ChooseREINFORCE, reinforce_update
    from recnn_amd.parallel import VocabParallelCritic, VocabParallelDiscreteActor
    S, N, H, B, steps, ps = 1290, 20_000, 256, 64, 12, 5
    torch.manual_seed(0)
    policy = recnn_amd.nn.DiscreteActor(S, N, H).to(dev)
    value = recnn_amd.nn.Critic(S, N, H, 54e-2).to(dev).eval()
    with torch.no_grad():
        value.linear3.weight.mul_(1e3)               # (the reference's 3e-5 init would make every TD target ~0)
    full = {"policy_net": policy, "target_policy_net": copy.deepcopy(policy), "value_net": value, "target_value_net": copy.deepcopy(value)}
    shard = {"policy_net": VocabParallelDiscreteActor.from_full(policy), "target_policy_net": VocabParallelDiscreteActor.from_full(policy),
             "value_net": VocabParallelCritic.from_full(value, S).eval(), "target_value_net": VocabParallelCritic.from_full(value, S).eval()}
    gen = torch.Generator().manual_seed(1)
    batches = []
    for _ in range(2):
        idx = torch.randint(0, N, (B,), generator=gen).to(dev)
        batches.append({"state": torch.randn(B, S, generator=gen).to(dev), "idx": idx, "reward": torch.randn(B, generator=gen).to(dev),
                        "next_state": torch.randn(B, S, generator=gen).to(dev), "done": (torch.rand(B, generator=gen) < 0.2).float().to(dev)})
    draws = torch.randint(0, N, (steps, B), generator=gen).to(dev)
    params = {"reinforce": ChooseREINFORCE(ChooseREINFORCE.basic_reinforce), "K": 10, "gamma": 0.99, "min_value": -10, "max_value": 10,
              "policy_step": ps, "soft_tau": 0.05}
    losses = {}
    for tag, nets in (("full", full), ("shard", shard)):
        opt = {"policy_optimizer": torch.optim.Adam(nets["policy_net"].parameters(), lr=1e-4),
               "value_optimizer": torch.optim.Adam(nets["value_net"].parameters(), lr=1e-4)}
        losses[tag] = []
        for t in range(steps):
            b = batches[t % 2]
            nets["policy_net"].forced_actions[:] = [draws[t]]
            batch = {"state": b["state"], "action": F_hip.onehot_rows(b["idx"], N), "reward": b["reward"], "next_state": b["next_state"],
                     "done": b["done"]}
            out = reinforce_update(batch, params, nets, opt, device=dev, step=t)
            if out is not None:
                losses[tag].append([out["value"], out["policy"]])
        torch.cuda.synchronize()

    def rel(a, b):
        return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))
    sp, sv = shard["policy_net"], shard["value_net"]
    n0, n1 = sp.n0, sp.n1
    res = {"rank": rank, "n0": n0, "n1": n1, "losses_full": losses["full"], "losses_shard": losses["shard"]}
    for tag in ("", "target_"):
        fp, fv, sp, sv = full[tag + "policy_net"], full[tag + "value_net"], shard[tag + "policy_net"], shard[tag + "value_net"]
        res[tag + "policy_w1"] = rel(sp.linear1.weight, fp.linear1.weight)
        res[tag + "policy_w2"] = rel(sp.linear2.weight, fp.linear2.weight[n0:n1])
        res[tag + "value_w1_state"] = rel(sv.linear1_state.weight, fv.linear1.weight[:, :S])
        res[tag + "value_w1_action"] = rel(sv.w1_action, fv.linear1.weight[:, S + n0:S + n1])
        res[tag + "value_w2"] = rel(sv.linear2.weight, fv.linear2.weight)
        res[tag + "value_w3"] = rel(sv.linear3.weight, fv.linear3.weight)
    every = [None] * world
    dist.all_gather_object(every, res)
    if rank == 0:
        os.makedirs(out_dir, exist_ok=True)
        json.dump(every, open(os.path.join(out_dir, "vp2_update.json"), "w"))


def main():
    out_dir = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo")
    if len(sys.argv) > 2 and sys.argv[2] == "update":
        run_update(out_dir, rank, world, dev)
        dist.barrier()
        dist.destroy_process_group()
        return
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
