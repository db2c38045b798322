"""REINFORCE (BASELINE configs[4]) at a 100k-item catalogue: per-step time of recnn_amd.nn.Reinforce.update.

usage: python tools/reinforce_bench.py [--items 100000] [--rows 256] [--hidden 2048] [--steps 31] [--method topk|basic]
                                       [--dtype fp32|bf16] [--beta learned|frozen] [--world N]
Prints one JSON line (median ms of ordinary steps, ms of policy-update steps, it/s over whole policy cycles).

Batches (round 5) come from a discrete-action `FrameEnv` -- a synthetic replay store whose item ids span the
catalogue, `embed_batch = batch_contstate_discaction` (recnn/data/utils.py:84-120: continuous state, ONE-HOT action over the
catalogue), `rows_per_batch` transition rows per batch, a new batch per step from `env.train_dataloader` -- not one random batch
repeated.

--world N > 1 (item 4b; also `python bench.py --algo reinforce --gpus N`): one process per GPU, the catalogue dimension SHARDED --
`VocabParallelDiscreteActor` (linear2 rows) and `VocabParallelCritic` (linear1 action columns) from recnn_amd/parallel.py inside the
same `reinforce_update` (recnn/nn/update/reinforce.py:81-129); every rank steps on the SAME batches (the replicated layers must see
the same inputs), the behaviour policy `Beta` is replicated.  Launched by torch.distributed.run (RANK / WORLD_SIZE in the
environment) or by itself (`--world N` without a launcher starts the ranks).  RECNN_BENCH_SINGLE_DEVICE=1 puts every rank on GPU 0 over
gloo: a functional check on a one-GPU box, not a measurement.  No multi-GPU node was available to the builder: the sharded form is
unmeasured."""
import argparse
import functools
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import recnn_amd  # noqa: E402
from recnn_amd.nn import functional as F_hip  # noqa: E402


def run(items=100000, rows=256, hidden=2048, steps=31, method="topk", optimizer="ranger", dtype="fp32", beta="learned", world=1):
    """One measurement (the body of `main`): returns the record as a dict."""
    a = argparse.Namespace(items=items, rows=rows, hidden=hidden, steps=steps, method=method, optimizer=optimizer, dtype=dtype, beta=beta,
                           world=world)
    return _measure(a)


def make_env(n_items, rows, device, seed=0, n_users=4096, frame=10):
    """Synthetic discrete-action env: users of 20..60 ratings over a catalogue of n_items (ids uniform: the worst case for reuse),
    ratings in the reference's 2 (r - 2.5) scale, embedding table randn(n_items, 128); batches of `rows` transition rows."""
    rng = np.random.default_rng(seed)
    lens = rng.integers(20, 61, size=n_users).astype(np.int64)
    off = np.zeros(n_users + 1, dtype=np.int64)
    off[1:] = np.cumsum(lens)
    total = int(off[-1])
    ids = rng.integers(0, n_items, size=total, dtype=np.int32)
    ratings = (2.0 * (rng.integers(1, 11, size=total) * 0.5 - 2.5)).astype(np.float32)
    table = torch.randn(n_items, 128, generator=torch.Generator().manual_seed(seed))
    users_per_batch = max(1, -(-rows // 10))          # >= rows windows whatever the lengths drawn (every user has >= 10)
    embed = functools.partial(recnn_amd.data.batch_contstate_discaction, num_items=n_items)
    return recnn_amd.data.env.FrameEnv.from_store(table, ids, ratings, off, frame_size=frame, batch_size=users_per_batch, device=device,
                                                  test_fraction=0.0, rows_per_batch=rows, embed_batch=embed)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--items", type=int, default=100000)
    ap.add_argument("--rows", type=int, default=256)
    ap.add_argument("--hidden", type=int, default=2048)
    ap.add_argument("--steps", type=int, default=31)
    ap.add_argument("--method", default="topk")
    ap.add_argument("--optimizer", default="ranger")
    ap.add_argument("--dtype", default="fp32", help="compute type of the catalogue GEMMs: fp32 | bf16")
    ap.add_argument("--beta", default="learned", choices=["learned", "frozen"],
                    help="behaviour policy of the Top-K correction: the notebook's Beta net trained inside every step (default) or a frozen projection")
    ap.add_argument("--world", type=int, default=int(os.environ.get("WORLD_SIZE", "1")),
                    help="ranks (one per GPU) the catalogue dimension is sharded over")
    a = ap.parse_args()
    if a.world > 1 and "RANK" not in os.environ:
        spawn(a.world, sys.argv[1:])                  # (does not return)
    rec = _measure(a)
    if int(os.environ.get("RANK", "0")) == 0:
        print(json.dumps(rec))


def spawn(world, argv):
    """Be the launcher: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... reinforce_bench.py`."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


def _measure(a):
    world = int(getattr(a, "world", 1) or 1)
    rank = int(os.environ.get("RANK", "0"))
    single = bool(os.environ.get("RECNN_BENCH_SINGLE_DEVICE"))
    local = 0 if single else int(os.environ.get("LOCAL_RANK", "0"))
    backend = None
    if world > 1:
        import torch.distributed as dist
        if int(os.environ.get("WORLD_SIZE", "1")) != world:
            raise SystemExit(f"--world {world} but the launcher started {os.environ.get('WORLD_SIZE', '1')} ranks")
        torch.cuda.set_device(local)
        backend = os.environ.get("RECNN_BENCH_BACKEND", "gloo" if single else "nccl")
        if not dist.is_initialized():
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", local))
            else:
                dist.init_process_group(backend)
    dev = torch.device("cuda", local)
    from recnn_amd._tune import apply_env_knobs
    apply_env_knobs()                                     # RECNN_* debug / tuning knobs for A/B runs from the shell
    recnn_amd.nn.algo.set_default_optimizer(a.optimizer)
    F_hip.set_catalogue_dtype(a.dtype)
    N, S, H, B = a.items, 1290, a.hidden, a.rows
    torch.manual_seed(0)                                  # the same replicated initial weights on every rank
    value = recnn_amd.nn.Critic(S, N, H, 54e-2).to(dev)
    policy = recnn_amd.nn.DiscreteActor(S, N, H).to(dev)
    if world > 1:
        from recnn_amd.parallel import VocabParallelCritic, VocabParallelDiscreteActor
        full_v, full_p = value, policy
        value = VocabParallelCritic.from_full(full_v, S)
        policy = VocabParallelDiscreteActor.from_full(full_p)
        del full_v, full_p
        torch.cuda.empty_cache()
    algo = recnn_amd.nn.Reinforce(policy, value).to(dev)
    beta_ms = []
    if a.method == "topk":
        if a.beta == "learned":
            # the notebook's configuration (3. TopK Reinforce Off Policy Correction.ipynb, cells 3-5): the behaviour policy is a
            # `Beta` net -- Linear(1290, n_items) + softmax -- that takes one optimizer step on its cross entropy inside EVERY call
            beta_net = recnn_amd.nn.Beta(S, N).to(dev)

            def beta(state, action=None):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                out = beta_net(state, action)
                e1.record()
                beta_ms.append((e0, e1))
                return out
        else:
            Wb = torch.randn(S, N, device=dev) * 0.02      # rounds 2-3: a FROZEN random projection (not the notebook's setup)

            def beta(state, action=None):
                return torch.softmax(state @ Wb, dim=1)
        policy.select_action = lambda state, action, K, writer, step, **kw: \
            policy._select_action_with_TopK_correction(state, beta, action, K=K, writer=writer, step=step)
        ch = recnn_amd.nn.ChooseREINFORCE
        algo.params["reinforce"] = ch(ch.reinforce_with_TopK_correction)
        policy.action_source = {"pi": "beta", "beta": "beta"}
    # ---- batches: a discrete-action FrameEnv, a new batch per step (same seed on every rank: the ranks of a sharded run must see
    # the same rows)
    env = make_env(N, B, dev, seed=0)
    torch.manual_seed(1234)                               # the loader's epoch permutation comes from the CPU generator
    stream = iter(env.train_dataloader)

    def next_batch():
        nonlocal stream
        try:
            b = next(stream)
        except StopIteration:
            stream = iter(env.train_dataloader)
            b = next(stream)
        assert b["state"].shape == (B, S) and b["action"].shape == (B, N)
        return b
    times, kinds, collate = [], [], []
    for t in range(a.steps):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        batch = next_batch()
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        out = algo.update(batch)
        algo.step()
        torch.cuda.synchronize(dev)
        t2 = time.perf_counter()
        collate.append(1e3 * (t1 - t0))
        times.append(1e3 * (t2 - t0))                    # a step = collate (gather + one-hot rows) + update
        kinds.append(out is not None)
    ordinary = sorted(x for x, k in zip(times[1:], kinds[1:]) if not k)
    pol = [x for x, k in zip(times, kinds) if k]
    cyc = times[11:31] if len(times) >= 31 else times[1:]
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([sum(cyc)], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        cyc_total = float(tt.item())
    else:
        cyc_total = sum(cyc)
    beta_step_ms = sorted(e0.elapsed_time(e1) for e0, e1 in beta_ms[1:])
    rec = {"n_items": N, "rows": B, "hidden": H, "method": a.method, "optimizer": a.optimizer, "dtype": a.dtype,
           "beta": a.beta if a.method == "topk" else None,
           "beta_train_call_ms": round(beta_step_ms[len(beta_step_ms) // 2], 3) if beta_step_ms else None,
           "ordinary_step_ms": round(ordinary[len(ordinary) // 2], 3) if ordinary else None, "policy_step_ms": [round(x, 2) for x in pol],
           "collate_ms": round(sorted(collate)[len(collate) // 2], 3),
           "it_per_s": round(1e3 * len(cyc) / cyc_total, 2), "peak_mem_gb": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2),
           "batches": "discrete-action FrameEnv (batch_contstate_discaction), a new batch per step, collate inside the timed step",
           "world": world}
    if world > 1:
        rec.update(sharding="catalogue dimension: VocabParallelDiscreteActor.linear2 rows + VocabParallelCritic.linear1 action columns; "
                            "Beta replicated; every rank steps on the same batches", backend=backend, rank=rank,
                   shard=[int(policy.n0), int(policy.n1)],
                   note="ranks share ONE GPU (RECNN_BENCH_SINGLE_DEVICE): functional, not a measurement" if single else
                        "first multi-GPU measurement of this path, if you are reading a real number")
        import torch.distributed as dist
        dist.barrier()
    return rec


if __name__ == "__main__":
    main()
