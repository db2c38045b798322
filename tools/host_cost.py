"""Host-side cost of one timed `run(20)` region (the driver's command): Python in front of the graph launch, hipGraphLaunch itself, the wait for
the GPU in read_losses, and the per-call cost of the context's checks.  usage: python tools/host_cost.py"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import recnn_amd
from recnn_amd.nn import fused
from recnn_amd import _lib as L
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
items, ratings, off, lens = bench.synthetic_store(0)
table = torch.randn(bench.N_ITEMS, bench.EMB, generator=torch.Generator().manual_seed(0))
env = recnn_amd.data.env.FrameEnv.from_store(table, items, ratings, off, frame_size=bench.FRAME, batch_size=25, device=dev, test_fraction=0.0)
fused.set_defaults(dtype="bf16", mask_mode="hash", seed=1234)
recnn_amd.nn.algo.set_default_optimizer("adam")
torch.manual_seed(0)
algo = recnn_amd.nn.DDPG(recnn_amd.nn.Actor(bench.STATE, bench.EMB, bench.HIDDEN, 6e-1), recnn_amd.nn.Critic(bench.STATE, bench.EMB, bench.HIDDEN, 54e-2)).to(dev)
algo.attach_env(env, rows_per_batch=2048, users_per_batch=None, shard=(0, 1))
eng = algo._fused_ctx.engine
marks = {}
orig_call = L.call
def call(name, *a):
    if name in ("recnn_engine_graph_run", "recnn_engine_read_losses"):
        marks[name + ":in"] = time.perf_counter()
        r = orig_call(name, *a)
        marks[name + ":out"] = time.perf_counter()
        return r
    return orig_call(name, *a)
L.call = call
import recnn_amd.nn.engine as E
E.L.call = call
ctx = algo._fused_ctx
acc = {}
def wrap(obj, name):
    f = getattr(obj, name)
    def g(*a, **k):
        t = time.perf_counter()
        r = f(*a, **k)
        acc.setdefault(name, []).append((time.perf_counter() - t) * 1e6)
        return r
    setattr(obj, name, g)
for n in ("ensure", "set_hyper", "apply_external", "run_steps", "_own_batch", "bump", "mark_stepped"):
    wrap(ctx, n)
for n in ("_fused_adam_cfgs", "flush", "_execute"):
    wrap(algo, n)
wrap(eng, "graph_run")
stream = torch.cuda.Stream(device=dev)
with torch.cuda.stream(stream):
    algo.run(5)
    for r in range(8):
        algo.prepare_run(20, first_step=5 + 20 * r)
    rec = []
    for r in range(8):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        algo.run(20)
        t4 = time.perf_counter()
        torch.cuda.synchronize(dev)
        t5 = time.perf_counter()
        rec.append([(marks["recnn_engine_graph_run:in"] - t0) * 1e6, (marks["recnn_engine_graph_run:out"] - marks["recnn_engine_graph_run:in"]) * 1e6,
                    (marks["recnn_engine_read_losses:in"] - marks["recnn_engine_graph_run:out"]) * 1e6,
                    (marks["recnn_engine_read_losses:out"] - marks["recnn_engine_read_losses:in"]) * 1e6, (t4 - marks["recnn_engine_read_losses:out"]) * 1e6,
                    (t5 - t4) * 1e6, (t5 - t0) * 1e6])
    rec = np.array(rec)[2:]
    print("us: python before graph launch | hipGraphLaunch (host) | python between | read_losses (waits for the GPU) | python after | final sync | region")
    print(np.round(np.median(rec, axis=0), 1))

print({k: round(float(np.median(v[3:])), 1) for k, v in acc.items()})
