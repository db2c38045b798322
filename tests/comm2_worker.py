"""Worker of tests/test_gpu_comm.py: launched W times by torch.distributed.run (gloo; all ranks share cuda:0 -- hipIpc maps
a buffer of the same device just as well as a peer's, so the whole protocol of csrc/comm.hip runs on a 1-GPU box).

  allreduce   recnn_dp_allreduce_flat against gloo's all-reduce of the same vectors: bit-equal at world 2 (a + b either way),
              within fp32 summation-order distance at world 3; repeated, odd sizes, and replayed from a captured graph
  stepper     DataParallelStepper with the device collective (run graphs / eager) against the same stepper on
              dist.all_reduce: parameters, optimizer state and losses bit-equal
"""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run_allreduce(out):
    from recnn_amd.parallel import PeerComm
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = torch.device("cuda", 0)
    comm = PeerComm.create(500_000)          # connects, then checks itself on known vectors; None on every rank if any failed
    assert comm is not None
    gen = torch.Generator().manual_seed(100 + rank)
    worst, reps = 0.0, 0
    for n in (1, 3, 4, 5, 255, 1024, 4099, 430_337, 500_000):
        for rep in range(6):
            x = torch.randn(n, generator=gen) * (1.0 + rep)
            want = x.clone()
            dist.all_reduce(want)                      # gloo, on the host
            got = comm.all_reduce(x.to(dev))
            torch.cuda.synchronize()
            d = float((got.cpu() - want).abs().max())
            worst = max(worst, d / float(want.abs().max()))
            if world == 2:
                assert torch.equal(got.cpu(), want), (n, rep, d)
            reps += 1
    # every rank holds the same bits whatever the world size
    x = torch.randn(77_777, generator=gen).to(dev)
    comm.all_reduce(x)
    ref = x.clone().cpu()
    dist.broadcast(ref, 0)
    assert torch.equal(x.cpu(), ref)
    # graph replay: three collectives per replay, epochs advance on the device
    a, b = torch.zeros(10_000, device=dev), torch.zeros(333, device=dev)
    side = torch.cuda.Stream(device=dev)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        comm.all_reduce(a); comm.all_reduce(b)        # warm-up outside capture
        side.synchronize()
        with torch.cuda.graph(g, stream=side):
            comm.all_reduce(a)
            comm.all_reduce(b)
            comm.all_reduce(a)
        for it in range(25):
            a.fill_(float(rank + 1)); b.fill_(float(it))
            g.replay()
            side.synchronize()
            s1 = world * (world + 1) / 2
            assert float(a[0]) == s1 * world and float(a[-1]) == s1 * world, (it, float(a[0]))
            assert float(b[7]) == it * world
    comm.check()
    dist.barrier()
    comm.close()
    out["allreduce"] = {"world": world, "collectives": reps, "worst_rel": worst}


def run_stepper(out, dtype, mode, small=False):
    from recnn_amd import _lib as L
    from recnn_amd.nn.engine import StepEngine
    from recnn_amd.parallel import DataParallelStepper, PeerComm
    from tests.dp2_worker import init_nets
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = torch.device("cuda", 0)
    S, A, H, B, steps, pe = 1290, 128, 256, 2048, 8, 3
    if small:
        # networks small enough that the critics' optimizer launch is a few dozen workgroups: both ranks' launches are resident
        # at once on the shared GPU, so the exchange can run INSIDE them (optim.hip exchange_grads, RECNN_COMM_FUSED=1)
        S, A, H, B = 34, 16, 32, 128
    Bl = B // world
    actor, critic = init_nets(0, S, A, H)
    gen = torch.Generator().manual_seed(1)
    batches, masks = [], []
    for _ in range(steps):
        batches.append({"state": torch.randn(B, S, generator=gen), "action": torch.randn(B, A, generator=gen),
                        "reward": torch.randn(B, generator=gen) * 3.0, "next_state": torch.randn(B, S, generator=gen),
                        "done": (torch.rand(B, generator=gen) < 0.1).float()})
        masks.append([(torch.rand(B, H, generator=gen) < 0.5).to(torch.uint8) for _ in range(6)])

    def make():
        eng = StepEngine("ddpg", S, A, H, Bl, dtype=dtype, mask_mode="external", seed=0, device=dev)
        for ni, p in ((L.NET_POLICY, actor), (L.NET_TARGET_POLICY, actor), (L.NET_VALUE1, critic), (L.NET_TARGET_VALUE1, critic)):
            eng.load_params(ni, p)
        eng.set_hyper(policy_opt=dict(lr=1e-4, weight_decay=1e-2), value_opt=dict(lr=1e-4, weight_decay=1e-2), policy_every=pe)
        eng.set_counters()
        return eng

    def drive(eng, dp):
        losses = []
        for t in range(steps):
            b = batches[t]
            lo, hi = rank * Bl, (rank + 1) * Bl
            eng.pack_batch(b["state"][lo:hi], b["action"][lo:hi], b["reward"][lo:hi], b["next_state"][lo:hi], b["done"][lo:hi])
            eng.set_external(masks=[m[lo:hi] for m in masks[t]])
            dp.step(t)
            lo_ = eng.losses()
            losses.append([lo_["value"], lo_["policy"]])
        torch.cuda.synchronize()
        return losses

    side = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(side):
        host = make()
        host_losses = drive(host, DataParallelStepper(host, Bl, use_graphs=(mode == "graphs")))
        # bf16: a region per network (gradients produced into / consumed from the peer buffer); fp32: one shared region, every
        # collective copies its arena in and out (both paths of recnn_engine_set_comm)
        # (the exchange inside the optimizer launch needs the per-network regions: `small` takes them in fp32 too)
        comm = PeerComm(PeerComm.floats_for(host) if (dtype == "bf16" or small) else max(int(g.numel()) for g in host.grads.values()))
        devc = make()
        dp = DataParallelStepper(devc, Bl, use_graphs=(mode == "graphs"), comm=comm)
        dev_losses = drive(devc, dp)
        gap = dp.check_replicas([devc.params[ni] for ni in (L.NET_POLICY, L.NET_VALUE1, L.NET_TARGET_POLICY, L.NET_TARGET_VALUE1)])
        comm.check()
    side.synchronize()
    diff = {}
    for name, ni in (("policy", L.NET_POLICY), ("value", L.NET_VALUE1), ("target_policy", L.NET_TARGET_POLICY),
                     ("target_value", L.NET_TARGET_VALUE1)):
        diff[name] = float((host.params[ni] - devc.params[ni]).abs().max())
    for name, ni in (("policy_m", L.NET_POLICY), ("value_m", L.NET_VALUE1)):
        diff[name] = float((host.adam_m[ni] - devc.adam_m[ni]).abs().max())
    out["stepper"] = {"world": world, "param_diff": diff, "host_losses": host_losses, "dev_losses": dev_losses, "replica_gap": gap,
                      "comm_fused": int(devc.tuning.comm_fused), "optimizer_workgroups_hint": int(sum((g.numel() + 1023) // 1024 for g in devc.grads.values()))}
    dist.barrier()
    devc.set_comm(None)
    comm.close()


def main():
    out_dir, what = sys.argv[1], sys.argv[2]
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    from recnn_amd._tune import apply_env_knobs
    apply_env_knobs()
    out = {}
    if what == "allreduce":
        run_allreduce(out)
    elif what == "stepper_small":
        run_stepper(out, sys.argv[3], sys.argv[4], small=True)
    else:
        run_stepper(out, sys.argv[3], sys.argv[4])
    if dist.get_rank() == 0:
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, "comm2.json"), "w") as f:
            json.dump(out, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
