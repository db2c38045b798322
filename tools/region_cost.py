"""What a timed `run(n)` region costs beyond its steps: regions of n = 20 / 40 / 60 steps from the same phase (step 5 mod 10), made-to-order
graphs, timed like bench.py (synchronize; run(n); synchronize) -> slope (us/step) and intercept (fixed us per region); and the idle cost of the
pieces of the fixed part.  usage: python tools/region_cost.py"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import recnn_amd
from recnn_amd.nn import fused

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
items, ratings, off, lens = bench.synthetic_store(0)
table = torch.randn(bench.N_ITEMS, bench.EMB, generator=torch.Generator().manual_seed(0))
env = recnn_amd.data.env.FrameEnv.from_store(table, items, ratings, off, frame_size=bench.FRAME, batch_size=25, device=dev, test_fraction=0.0)
fused.set_defaults(dtype=os.environ.get("DTYPE", "bf16"), mask_mode="hash", seed=1234)
recnn_amd.nn.algo.set_default_optimizer("adam")
torch.manual_seed(0)
algo = recnn_amd.nn.DDPG(recnn_amd.nn.Actor(bench.STATE, bench.EMB, bench.HIDDEN, 6e-1), recnn_amd.nn.Critic(bench.STATE, bench.EMB, bench.HIDDEN, 54e-2)).to(dev)
algo.attach_env(env, rows_per_batch=2048, users_per_batch=None, shard=(0, 1))
eng = algo._fused_ctx.engine
stream = torch.cuda.Stream(device=dev)
res = {}
with torch.cuda.stream(stream):
    algo.run(5)
    for n in (20, 40, 60):
        algo.prepare_run(n, first_step=5)
    torch.cuda.synchronize(dev)
    for n in (20, 40, 60, 20, 40, 60):
        ts = []
        for r in range(9):
            pad = (5 - (algo._step % 10)) % 10          # back to phase 5
            if pad:
                algo.run(pad)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            algo.run(n)
            torch.cuda.synchronize(dev)
            ts.append((time.perf_counter() - t0) * 1e6)
        res.setdefault(n, []).append(float(np.median(ts)))
    for n, v in res.items():
        print(f"run({n}) from phase 5: median region {[round(x, 1) for x in v]} us -> {[round(x / n, 2) for x in v]} us/step")
    a = np.array([[n, 1.0] for n in res for _ in res[n]]); b = np.array([x for n in res for x in res[n]])
    slope, icpt = np.linalg.lstsq(a, b, rcond=None)[0]
    print(f"fit: {slope:.2f} us/step + {icpt:.1f} us per region")

    def t_of(f, k=200):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(k):
            f()
        return (time.perf_counter() - t0) / k * 1e6
    print(f"idle costs: torch.cuda.synchronize {t_of(lambda: torch.cuda.synchronize(dev)):.1f} us; engine.losses() (stream sync + pinned mirror) {t_of(eng.losses):.1f} us; "
          f"algo.flush() {t_of(algo.flush):.1f} us")
