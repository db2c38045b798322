"""cProfile of the planned-batch update() loop (where does the host time of a flush go?).  usage: python tools/loop_profile.py"""
import cProfile, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import recnn_amd
from recnn_amd.nn import fused

dev = torch.device("cuda", 0)
items, ratings, off, lens = bench.synthetic_store(0)
table = torch.randn(bench.N_ITEMS, bench.EMB, generator=torch.Generator().manual_seed(0))
env = recnn_amd.data.env.FrameEnv.from_store(table, items, ratings, off, frame_size=10, batch_size=25, device=dev, test_fraction=0.0,
                                             rows_per_batch=2048)
fused.set_defaults(dtype="bf16", mask_mode="hash", seed=1)
recnn_amd.nn.algo.set_default_optimizer("adam")
torch.manual_seed(0)
algo = recnn_amd.nn.DDPG(recnn_amd.nn.Actor(1290, 128, 256, 6e-1), recnn_amd.nn.Critic(1290, 128, 256, 54e-2)).to(dev)
algo.attach_env(env, rows_per_batch=2048, users_per_batch=256)


def loop(n):
    for batch in algo.batches(n):
        algo.update(batch, learn=True); algo.step()
    algo.flush()


stream = torch.cuda.Stream(device=dev) if len(sys.argv) > 1 and sys.argv[1] == "stream" else None
with (torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream())):
    loop(600)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    loop(1800)
    pr.disable()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
print(f"host {1e6 * (t1 - t0) / 1800:.1f} us/step, with the final sync {1e6 * (t2 - t0) / 1800:.1f} us/step")
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
