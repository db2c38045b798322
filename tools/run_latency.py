"""Host-visible cost of short Algo.run(n) calls at the bench shape: per-call wall time for a sequence of calls
(first use of each run-graph shape vs repeats).  usage: python tools/run_latency.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import recnn_amd
from recnn_amd.nn import fused

dev = torch.device("cuda", 0)
items, ratings, off, lens = bench.synthetic_store(0)
table = torch.randn(bench.N_ITEMS, bench.EMB, generator=torch.Generator().manual_seed(0))
env = recnn_amd.data.env.FrameEnv.from_store(table, items, ratings, off, frame_size=10, batch_size=25, device=dev, test_fraction=0.0)
fused.set_defaults(dtype="bf16", mask_mode="hash", seed=1)
torch.manual_seed(0)
algo = recnn_amd.nn.DDPG(recnn_amd.nn.Actor(1290, 128, 256, 6e-1), recnn_amd.nn.Critic(1290, 128, 256, 54e-2)).to(dev)
algo.attach_env(env, rows_per_batch=2048, users_per_batch=256)
stream = torch.cuda.Stream(device=dev)
with torch.cuda.stream(stream):
    for n in (5, 20, 20, 20, 20, 60, 60, 7, 7, 200, 200, 2000):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        algo.run(n)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"run({n:5d}) from step {algo._step - n:5d}: {dt * 1e6:9.1f} us total, {dt * 1e6 / n:7.2f} us/step", flush=True)

# ---- where the fixed cost of a short run() goes: host-side pieces timed separately (no device work in "pre")
ctx = algo._fused_ctx
import time as _t
with torch.cuda.stream(stream):
    for n in (20, 20, 20):
        torch.cuda.synchronize()
        t0 = _t.perf_counter()
        cfgs = algo._fused_adam_cfgs(algo._fused_keys)
        ctx.ensure(algo.nets, ctx.sampler["rows"])
        ctx.set_hyper(algo.params, cfgs[0], cfgs[1])
        ctx.apply_external(ctx.sampler["rows"])
        t1 = _t.perf_counter()
        ctx.run_steps(algo._step, n)
        t2 = _t.perf_counter()
        algo._step += n
        lo = ctx.engine.losses()
        t3 = _t.perf_counter()
        print(f"pieces of run({n}): python pre {1e6 * (t1 - t0):7.1f} us | graph launches (host) {1e6 * (t2 - t1):7.1f} us | "
              f"wait + loss read-back {1e6 * (t3 - t2):7.1f} us | total {1e6 * (t3 - t0):7.1f}", flush=True)
