#!/usr/bin/env python3
"""multigpu_preflight.py -- make the FIRST run on a multi-GPU node diagnostic.

No node with more than one GPU was available while this repository was built: neither the device-side gradient exchange
(csrc/comm.hip, hipIpc peer buffers over xGMI) nor the RCCL fallback has ever crossed a link.  This script walks the N > 1 path stage
by stage and prints ONE JSON object per stage (rank 0, flushed, a "begin" line before every stage), so that a hang or a time-out is
attributed to a stage instead of to "bench rc != 0":

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P tools/multigpu_preflight.py
    python bench.py --gpus N --preflight                      (starts the ranks itself)
    RECNN_BENCH_SINGLE_DEVICE=1 python bench.py --gpus 2 --preflight      (functional check on a 1-GPU box: ranks share GPU 0, gloo)

Stages: devices -> process_group (backend init, barrier, one RCCL all-reduce) -> peer_connect (hipIpc export / map, waits bounded by
200 ms) -> peer_self_test -> collective_latency (peer vs RCCL, four sizes) -> identity (N ranks x B/N rows == 1 rank x B rows with the
real engine, fp32) -> replicas (bit-identical parameters, no broadcast) -> bench_peer (5 + 20 steps through the run graphs with the
collectives inside) -> bench_rccl (the same with host-issued RCCL all-reduces between phase graphs).
"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

S, A, H, ROWS = 1290, 128, 256, 2048


def emit(rank, stage, **kw):
    if rank == 0:
        print(json.dumps({"stage": stage, **kw}), flush=True)


def init_nets(seed):
    torch.manual_seed(seed)

    def mk(inp, out, init_w):
        l1, l2, l3 = torch.nn.Linear(inp, H), torch.nn.Linear(H, H), torch.nn.Linear(H, out)
        l3.weight.data.uniform_(-init_w, init_w)
        l3.bias.data.uniform_(-init_w, init_w)
        return {"w1": l1.weight.data.clone(), "b1": l1.bias.data.clone(), "w2": l2.weight.data.clone(), "b2": l2.bias.data.clone(),
                "w3": l3.weight.data.clone(), "b3": l3.bias.data.clone()}
    critic = mk(S + A, 1, 54e-2)
    actor = mk(S, A, 6e-1)
    return actor, critic


def make_engine(L, dev, rows, dtype, mask_mode, seed=0):
    from recnn_amd.nn.engine import StepEngine
    actor, critic = init_nets(0)
    eng = StepEngine("ddpg", S, A, H, rows, dtype=dtype, mask_mode=mask_mode, seed=seed, device=dev)
    for ni, p in ((L.NET_POLICY, actor), (L.NET_TARGET_POLICY, actor), (L.NET_VALUE1, critic), (L.NET_TARGET_VALUE1, critic)):
        eng.load_params(ni, p)
    eng.set_hyper(policy_opt=dict(lr=1e-5, weight_decay=1e-2), value_opt=dict(lr=1e-5, weight_decay=1e-2), policy_every=10)
    eng.set_counters()
    return eng


LAST_SUMMARY = None     # {"ok": bool, "stages": {stage: ok}} of the latest main() in this process (bench.py embeds it)


def main(keep_group=False, quick=False):
    """keep_group: leave the process group initialised (bench.py runs this in front of its timed regions and goes on with the group).
    quick: stop after the replica check -- no latency table, no 25-step runs of the two data-parallel paths (bench.py does those itself)."""
    global LAST_SUMMARY
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    single = bool(os.environ.get("RECNN_BENCH_SINGLE_DEVICE"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if single:
        local_rank = 0
        os.environ.setdefault("RECNN_COMM_WORKGROUPS", "32")
        os.environ.setdefault("RECNN_COMM_FUSED", "0")
    summary = {}

    def stage(name, fn):
        emit(rank, name, event="begin")
        t0 = time.perf_counter()
        try:
            out = fn() or {}
            ok = bool(out.pop("ok", True))
        except Exception as ex:          # a stage that fails is reported; later stages that can still run do
            out, ok = {"error": f"{type(ex).__name__}: {ex}"[:400]}, False
        summary[name] = ok
        emit(rank, name, event="end", ok=ok, seconds=round(time.perf_counter() - t0, 3), **out)
        return ok

    # ---------------------------------------------------------------- devices
    def st_devices():
        n = torch.cuda.device_count()
        torch.cuda.set_device(local_rank)
        return {"ok": n > local_rank, "world": world, "visible_devices": n, "device": torch.cuda.get_device_name(local_rank),
                "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"), "ranks_share_one_gpu": single}
    if not stage("devices", st_devices):
        return 2
    dev = torch.device("cuda", local_rank)
    from recnn_amd import _lib as L
    from recnn_amd._tune import apply_env_knobs
    apply_env_knobs()

    # ---------------------------------------------------------------- process group
    def st_pg():
        backend = os.environ.get("RECNN_BENCH_BACKEND", "gloo" if single else "nccl")
        if not dist.is_initialized():
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=dev)
            else:
                dist.init_process_group(backend)
        dist.barrier()
        x = torch.full((1024,), float(rank + 1), device=dev if backend == "nccl" else "cpu")
        t0 = time.perf_counter()
        dist.all_reduce(x)
        if backend == "nccl":
            torch.cuda.synchronize(dev)
        return {"ok": float(x[0]) == world * (world + 1) / 2, "backend": backend + (" (= RCCL)" if backend == "nccl" else ""),
                "first_all_reduce_ms": round((time.perf_counter() - t0) * 1e3, 3)}
    if not stage("process_group", st_pg):
        return 3
    backend_gpu = dist.get_backend() == "nccl"

    def rccl_all_reduce(t):
        if backend_gpu:
            dist.all_reduce(t)
        else:                           # gloo stand-in on a 1-GPU box: through the host
            h = t.cpu()
            dist.all_reduce(h)
            t.copy_(h)

    # ---------------------------------------------------------------- peer communicator
    from recnn_amd.parallel import DataParallelStepper, PeerComm
    probe = make_engine(L, dev, 64, "fp32", "none")
    floats = PeerComm.floats_for(probe)
    del probe
    state = {"comm": None}

    def st_connect():
        comm = PeerComm(floats)          # hipIpc export, all_gather of the handles, map every peer
        comm.set_timeout_ms(200)         # first contact: a peer that does not answer costs 0.2 s per wait, not 4
        state["comm"] = comm
        return {"floats": floats, "world": comm.world}
    have_peer = stage("peer_connect", st_connect)
    # AGREE on the outcome before any stage that only some ranks would enter: a connect that failed on ONE rank must
    # take every rank down the same path -- the self-test below is a collective
    votes = [None] * world
    dist.all_gather_object(votes, int(have_peer))
    if have_peer and not min(votes):
        summary["peer_connect"] = False
        emit(rank, "peer_connect", event="vote", ok=False, per_rank=votes)
    have_peer = bool(min(votes))

    def st_selftest():
        ok = state["comm"].self_test()
        votes = [None] * world
        dist.all_gather_object(votes, int(ok))
        return {"ok": min(votes) == 1, "per_rank": votes}
    if have_peer:
        have_peer = stage("peer_self_test", st_selftest)
    if not have_peer and state["comm"] is not None:
        try:
            state["comm"].clear_status()
        except Exception:
            pass

    # ---------------------------------------------------------------- one collective per size class
    def st_latency():
        out = {}
        for n in (1024, 65536, 429_313, floats):
            x = torch.randn(n, device=dev)
            row = {}
            for name, fn in (("peer", state["comm"].all_reduce if have_peer else None), ("rccl", rccl_all_reduce)):
                if fn is None:
                    continue
                for _ in range(3):
                    fn(x)
                torch.cuda.synchronize(dev)
                dist.barrier()
                t0 = time.perf_counter()
                for _ in range(20):
                    fn(x)
                torch.cuda.synchronize(dev)
                row[name + "_us"] = round((time.perf_counter() - t0) / 20 * 1e6, 2)
            out[str(n)] = row
        if have_peer:
            state["comm"].check()
        return {"floats": out, "note": "host-timed, 20 back-to-back collectives, includes launch overhead"}
    if not quick:
        stage("collective_latency", st_latency)

    # ---------------------------------------------------------------- N ranks x B/N rows == 1 rank x B rows (real engine, fp32)
    gen = torch.Generator().manual_seed(7)
    steps = 3
    batches = [{"state": torch.randn(ROWS, S, generator=gen), "action": torch.randn(ROWS, A, generator=gen),
                "reward": torch.randn(ROWS, generator=gen) * 3.0, "next_state": torch.randn(ROWS, S, generator=gen),
                "done": (torch.rand(ROWS, generator=gen) < 0.1).float()} for _ in range(steps)]
    masks = [[(torch.rand(ROWS, H, generator=gen) < 0.5).to(torch.uint8) for _ in range(6)] for _ in range(steps)]
    bl = ROWS // world

    def drive(eng, step_fn, lo, hi):
        losses = []
        for t in range(steps):
            b = batches[t]
            eng.pack_batch(b["state"][lo:hi], b["action"][lo:hi], b["reward"][lo:hi], b["next_state"][lo:hi], b["done"][lo:hi])
            eng.set_external(masks=[m[lo:hi] for m in masks[t]])
            step_fn(t)
            losses.append(eng.losses())
        return losses

    def st_identity():
        if ROWS % world:
            return {"ok": False, "error": f"{ROWS} rows do not split over {world} ranks"}
        ref = make_engine(L, dev, ROWS, "fp32", "external")
        ref_losses = drive(ref, lambda t: ref.step(ROWS, True, t), 0, ROWS)
        out = {}
        worst = 0.0
        for name in (("peer",) if have_peer else ()) + ("rccl",):
            eng = make_engine(L, dev, bl, "fp32", "external")
            dp = DataParallelStepper(eng, bl, use_graphs=False, comm=state["comm"] if name == "peer" else None)
            got = drive(eng, dp.step, rank * bl, (rank + 1) * bl)
            # the loss of the global batch is the mean of the ranks' losses (equal slices)
            dev_l = 0.0
            for t in range(steps):
                v = torch.tensor([got[t]["value"], got[t]["policy"]], dtype=torch.float64)
                dist.all_reduce(v)
                v /= world
                for k, x in zip(("value", "policy"), v.tolist()):
                    dev_l = max(dev_l, abs(x - ref_losses[t][k]) / (abs(ref_losses[t][k]) + 1e-6))
            pdiff = max(float((eng.params[ni] - ref.params[ni]).abs().max() / (ref.params[ni].abs().max() + 1e-30))
                        for ni in (L.NET_POLICY, L.NET_VALUE1))
            gap = dp.check_replicas([eng.params[ni] for ni in (L.NET_POLICY, L.NET_VALUE1, L.NET_TARGET_POLICY, L.NET_TARGET_VALUE1)])
            out[name] = {"worst_rel_loss_dev": dev_l, "param_rel_dev": pdiff, "replica_gap": gap}
            worst = max(worst, dev_l)
            summary.setdefault("_gaps", {})[name] = gap
            if name == "peer":
                eng.set_comm(None)
        return {"ok": worst <= 1e-4, "paths": out, "criterion": "losses of N x B/N within 1e-4 of 1 x B (fp32 summation order)"}
    stage("identity", st_identity)

    def st_replicas():
        gaps = summary.get("_gaps", {})
        return {"ok": bool(gaps) and all(g == 0.0 for g in gaps.values()), "replica_gap": gaps,
                "criterion": "parameters bit-identical on every rank without a broadcast"}
    stage("replicas", st_replicas)
    summary.pop("_gaps", None)

    if quick:
        emit(rank, "summary", ok=all(v for v in summary.values()), stages=summary, quick=True)
        LAST_SUMMARY = {"ok": all(v for v in summary.values()), "stages": dict(summary), "quick": True}
        dist.barrier()
        if state["comm"] is not None:
            state["comm"].close()
        if not keep_group:
            dist.destroy_process_group()
        return 0 if all(summary.values()) else 1
    # ---------------------------------------------------------------- 5 + 20 steps of the benchmark's data-parallel paths
    import bench as B
    items, ratings, off, lens = B.synthetic_store(0)
    table = torch.randn(B.N_ITEMS, B.EMB, generator=torch.Generator().manual_seed(0))
    import recnn_amd
    env = recnn_amd.data.env.FrameEnv.from_store(table, items, ratings, off, frame_size=B.FRAME, batch_size=25, device=dev, test_fraction=0.0)

    def bench_path(use_peer):
        algo = B._make_algo(recnn_amd, "ddpg", "bf16", dev, 1234 + rank)
        torch.manual_seed(100 + rank)
        algo.attach_env(env, rows_per_batch=ROWS, shard=(rank, world))
        eng = algo._fused_ctx.engine
        stream = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(stream):
            comm = state["comm"] if use_peer else None
            dp = DataParallelStepper(eng, ROWS, comm=comm)
            one_by_one = single and comm is not None          # ranks sharing one GPU are time-sliced: one step per host sync
            def run(first, n):
                if one_by_one:
                    for t in range(first, first + n):
                        dp.step(t)
                        torch.cuda.synchronize(dev)
                else:
                    dp.run(first, n)
            run(0, 5)
            torch.cuda.synchronize(dev)
            if comm is not None:
                comm.check()
                comm.set_timeout_ms(4000)                     # the warm-up made contact: back to the training bound
                eng.graph_build(ROWS)                         # (the bound travels in the captured kernel arguments)
            dist.barrier()
            t0 = time.perf_counter()
            run(5, 20)
            torch.cuda.synchronize(dev)
            dist.barrier()
            dt = time.perf_counter() - t0
            if comm is not None:
                comm.check()
                eng.set_comm(None)
        losses = eng.losses()
        return {"ok": all(abs(v) < 1e6 for v in losses.values()), "ms_per_step": round(dt / 20 * 1e3, 4), "steps_per_s_all_ranks": round(world * 20 / dt, 1),
                "final_losses": losses, "note": "ranks share one GPU: functional only" if single else ""}
    if have_peer:
        stage("bench_peer", lambda: bench_path(True))
    stage("bench_rccl", lambda: bench_path(False))

    emit(rank, "summary", ok=all(v for v in summary.values()), stages=summary)
    LAST_SUMMARY = {"ok": all(v for v in summary.values()), "stages": dict(summary)}
    dist.barrier()
    if state["comm"] is not None:
        state["comm"].close()
    if not keep_group:
        dist.destroy_process_group()
    return 0 if all(summary.values()) else 1


if __name__ == "__main__":
    sys.exit(main())
