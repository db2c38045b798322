"""Cost of an epoch boundary of Algo.run at the bench shape: _reshuffle alone (host + device, synchronised), and run(n) calls
that do / do not cross a boundary.  usage: python tools/reshuffle_cost.py"""
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
ctx = algo._fused_ctx
nb = ctx.sampler["n_batches"]
print("batches per epoch:", nb)
stream = torch.cuda.Stream(device=dev)
with torch.cuda.stream(stream):
    algo.run(50)
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ctx._reshuffle()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"_reshuffle: host {1e6 * (t1 - t0):8.1f} us, + device drain {1e6 * (t2 - t1):8.1f} us")
    # runs of 300 steps: with ~540 batches per epoch every second one crosses a boundary
    for i in range(8):
        cur = ctx.sampler["cursor"]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        algo.run(300)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"run(300) from cursor {cur:4d} ({'crosses' if cur + 300 >= nb else 'inside '}): {dt * 1e3:8.2f} ms, {dt * 1e6 / 300:7.2f} us/step")
    # one long call vs the same steps as short calls (is a long run() slower per step than its pieces?)
    for rep in range(3):
        for n, k in ((2000, 1), (200, 10), (540, 4), (4000, 1)):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(k):
                algo.run(n)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print(f"rep {rep}: {k:2d} x run({n:4d}): {dt * 1e3:8.2f} ms, {dt * 1e6 / (n * k):7.2f} us/step")
