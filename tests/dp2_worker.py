"""Worker of tests/test_gpu_parity_r2.py::test_two_ranks_on_one_gpu_equal_one_rank: launched twice by torch.distributed.run
(gloo; both ranks share cuda:0).  Every rank drives the REAL HIP engine through DataParallelStepper on its half of a
global batch (rows r*B/2 .. (r+1)*B/2, the matching slices of the global dropout masks); rank 0 then runs the same steps
on one engine with all B rows and writes the comparison to <out>/dp2_<dtype>.json.

N ranks x B/N rows == 1 rank x B rows up to fp32 summation order; the L1 clip acts on the REDUCED actor
gradient; replicas stay identical without a broadcast.
"""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def init_nets(seed, S, A, H):
    torch.manual_seed(seed)

    def mk(inp, out, init_w):
        l1, l2, l3 = torch.nn.Linear(inp, H), torch.nn.Linear(H, H), torch.nn.Linear(H, out)
        l3.weight.data.uniform_(-init_w, init_w)
        l3.bias.data.uniform_(-init_w, init_w)
        return {"w1": l1.weight.data.clone(), "b1": l1.bias.data.clone(), "w2": l2.weight.data.clone(),
                "b2": l2.bias.data.clone(), "w3": l3.weight.data.clone(), "b3": l3.bias.data.clone()}
    critic = mk(S + A, 1, 54e-2)
    actor = mk(S, A, 6e-1)
    return actor, critic


def main():
    out_dir, dtype, mode = sys.argv[1], sys.argv[2], sys.argv[3]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo")
    from recnn_amd import _lib as L
    from recnn_amd.nn.engine import StepEngine
    from recnn_amd.parallel import DataParallelStepper

    S, A, H, B, steps, pe = 1290, 128, 256, 2048, 7, 3
    Bl = B // world
    actor, critic = init_nets(0, S, A, H)
    gen = torch.Generator().manual_seed(1)
    batches, masks = [], []
    for _ in range(steps):
        batches.append({"state": torch.randn(B, S, generator=gen), "action": torch.randn(B, A, generator=gen),
                        "reward": torch.randn(B, generator=gen) * 3.0, "next_state": torch.randn(B, S, generator=gen),
                        "done": (torch.rand(B, generator=gen) < 0.1).float()})
        masks.append([(torch.rand(B, H, generator=gen) < 0.5).to(torch.uint8) for _ in range(6)])

    def make(rows):
        eng = StepEngine("ddpg", S, A, H, rows, dtype=dtype, mask_mode="external", seed=0, device=dev)
        for ni, p in ((L.NET_POLICY, actor), (L.NET_TARGET_POLICY, actor), (L.NET_VALUE1, critic), (L.NET_TARGET_VALUE1, critic)):
            eng.load_params(ni, p)
        eng.set_hyper(policy_opt=dict(lr=1e-4, weight_decay=1e-2), value_opt=dict(lr=1e-4, weight_decay=1e-2),
                      policy_every=pe)   # weight decay on both: no element sits in Adam's eps regime (|g| >= wd |p|)
        eng.set_counters()
        return eng

    def load(eng, t, lo, hi):
        b = batches[t]
        eng.pack_batch(b["state"][lo:hi], b["action"][lo:hi], b["reward"][lo:hi], b["next_state"][lo:hi], b["done"][lo:hi])
        eng.set_external(masks=[m[lo:hi] for m in masks[t]])

    # ---- data-parallel run: this rank's slice
    eng = make(Bl)
    side = torch.cuda.Stream(device=dev)
    dp_losses = []
    with torch.cuda.stream(side):
        dp = DataParallelStepper(eng, Bl, use_graphs=(mode == "graphs"))
        for t in range(steps):
            load(eng, t, rank * Bl, (rank + 1) * Bl)
            dp.step(t)
            lo = eng.losses()
            v = torch.tensor([lo["value"], lo["policy"]], dtype=torch.float64)
            dist.all_reduce(v)
            dp_losses.append((v / world).tolist())      # mean of equal-sized shards = the global batch mean
        replica_gap = dp.check_replicas([eng.params[ni] for ni in (L.NET_POLICY, L.NET_VALUE1, L.NET_TARGET_POLICY,
                                                                   L.NET_TARGET_VALUE1)])
    side.synchronize()
    torch.cuda.synchronize()
    dist.barrier()
    if rank == 0:
        # ---- the same steps on ONE engine with all B rows
        ref = make(B)
        ref_losses = []
        for t in range(steps):
            load(ref, t, 0, B)
            ref.step(B, True, t)
            lo = ref.losses()
            ref_losses.append([lo["value"], lo["policy"]])
        torch.cuda.synchronize()
        perr = {}
        for name, ni in (("policy", L.NET_POLICY), ("value", L.NET_VALUE1), ("target_policy", L.NET_TARGET_POLICY),
                         ("target_value", L.NET_TARGET_VALUE1)):
            a, b = eng.params[ni].double().cpu(), ref.params[ni].double().cpu()
            perr[name] = {"fro": float((a - b).norm() / b.norm()), "max": float((a - b).abs().max() / b.abs().max())}
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, f"dp2_{dtype}_{mode}.json"), "w") as f:
            json.dump({"dp_losses": dp_losses, "ref_losses": ref_losses, "param_err": perr, "replica_gap": replica_gap,
                       "world": world, "rows_per_rank": Bl}, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
