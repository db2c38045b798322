"""Where the fixed cost of a short timed region goes: cProfile + wall clock of `sync; algo.run(20); sync` repeated, as bench.py's
driver configuration (--steps 20 --warmup 5) times it.  usage: python tools/run20_profile.py"""
import cProfile, os, pstats, sys, time
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
recnn_amd.nn.algo.set_default_optimizer("adam")
torch.manual_seed(0)
algo = recnn_amd.nn.DDPG(recnn_amd.nn.Actor(1290, 128, 256, 6e-1), recnn_amd.nn.Critic(1290, 128, 256, 54e-2)).to(dev)
algo.attach_env(env, rows_per_batch=2048, users_per_batch=256)
stream = torch.cuda.Stream(device=dev)
N = 20
with torch.cuda.stream(stream):
    for r in range(12):
        algo.prepare_run(N, first_step=5 + r * N)
    algo.run(5)
    torch.cuda.synchronize()
    times = []
    pr = cProfile.Profile()
    for r in range(12):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if r >= 2:
            pr.enable()
        algo.run(N)
        if r >= 2:
            pr.disable()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        times.append((t1 - t0, t2 - t0))
    # the same 20 steps inside a long run: the marginal cost
    torch.cuda.synchronize(); t0 = time.perf_counter(); algo.run(2000); torch.cuda.synchronize(); long = (time.perf_counter() - t0) / 2000
print("run(20): host-return / synced, us:", [(round(a * 1e6), round(b * 1e6)) for a, b in times])
print(f"per step inside run(2000): {long * 1e6:.1f} us -> 20 steps = {20 * long * 1e6:.0f} us")
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)

# split launches: a short head graph gets the GPU going while the host is still enqueueing the long tail graph
ctx = algo._fused_ctx
every = algo.params["policy_step"]
with torch.cuda.stream(stream):
    for split in (0, 1, 2, 3, 4, 6):
        res = []
        for r in range(14):
            first = algo._step
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            if split:
                ctx.run_steps(first, split, every=every)
                ctx.run_steps(first + split, N - split, every=every)
            else:
                ctx.run_steps(first, N, every=every)
            torch.cuda.synchronize()
            res.append(time.perf_counter() - t0)
            algo._step += N
        res = sorted(res[4:])
        print(f"split {split}: median {1e6 * res[len(res) // 2]:.0f} us, min {1e6 * res[0]:.0f} us per {N} steps")

# the pieces of Algo.run(20), host time each (GPU idle before the launch; the launch returns before the GPU is done)
import collections
acc = collections.OrderedDict()
def tick(name, t0):
    t1 = time.perf_counter(); acc[name] = acc.get(name, 0.0) + (t1 - t0); return t1
keys = algo._fused_keys
with torch.cuda.stream(stream):
    for r in range(10):
        torch.cuda.synchronize()
        first = algo._step
        t = time.perf_counter()
        algo.flush(); t = tick("flush()", t)
        cfgs = algo._fused_adam_cfgs(keys); t = tick("_fused_adam_cfgs", t)
        ctx.ensure(algo.nets, ctx.sampler["rows"]); t = tick("ensure", t)
        ctx.set_hyper(algo.params, cfgs[0], cfgs[1]); t = tick("set_hyper", t)
        ctx.apply_external(ctx.sampler["rows"]); t = tick("apply_external", t)
        ctx.run_steps(first, N, every=every); t = tick("run_steps (graph launch)", t)
        for k, ni in zip(keys, (fused.L.NET_POLICY, fused.L.NET_VALUE1, fused.L.NET_VALUE2)):
            ctx.bump(algo.optimizers[k], ni, 2 if ni == fused.L.NET_POLICY else N)
        t = tick("bump", t)
        ctx.mark_stepped(list(ctx.modules)); t = tick("mark_stepped", t)
        ctx.engine.losses(); t = tick("losses (waits for the GPU)", t)
        algo._step += N
for k, v in acc.items():
    print(f"   {k:32s} {1e6 * v / 10:8.1f} us")
