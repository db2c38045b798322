"""Throughput of the reference-shaped loop `for batch in loader: algo.update(batch); algo.step()` (one host sync per step, like the
reference's .item() calls) at the bench shape, next to Algo.run.  usage: python tools/update_loop_rate.py [bf16|fp32]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import recnn_amd
from recnn_amd.nn import fused

dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
dev = torch.device("cuda", 0)
items, ratings, off, lens = bench.synthetic_store(0)
table = torch.randn(bench.N_ITEMS, bench.EMB, generator=torch.Generator().manual_seed(0))
env = recnn_amd.data.env.FrameEnv.from_store(table, items, ratings, off, frame_size=10, batch_size=25, device=dev, test_fraction=0.0,
                                             rows_per_batch=2048)
fused.set_defaults(dtype=dtype, mask_mode="hash", seed=1)
recnn_amd.nn.algo.set_default_optimizer("adam")
torch.manual_seed(0)
algo = recnn_amd.nn.DDPG(recnn_amd.nn.Actor(1290, 128, 256, 6e-1), recnn_amd.nn.Critic(1290, 128, 256, 54e-2)).to(dev)
users = env.base.train_user_dataset.users
batches = [env.collate_users([int(u) for u in users[i * 256:(i + 1) * 256]]) for i in range(8)]
for i in range(30):
    algo.update(batches[i % 8], learn=True); algo.step()
torch.cuda.synchronize()
n = 400
t0 = time.perf_counter()
for i in range(n):
    algo.update(batches[i % 8], learn=True); algo.step()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"update() loop, {dtype}, 2048 rows, batches resident: {n / dt:8.0f} steps/s, {dt / n * 1e6:7.1f} us/step")
t0 = time.perf_counter()
for i in range(n):
    b = env.collate_users([int(u) for u in users[(i % 8) * 256:(i % 8 + 1) * 256]])
    algo.update(b, learn=True); algo.step()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"collate + update() loop, {dtype}:                        {n / dt:8.0f} steps/s, {dt / n * 1e6:7.1f} us/step")

# the same loop shape on planned batches (Algo.batches): update() queues, the queue replays as 60-step run graphs
torch.manual_seed(1)
algo.attach_env(env, rows_per_batch=2048, users_per_batch=256)
for name, n in (("warm-up", 600), ("timed", 6000)):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    log = []
    t_flush, orig_flush = [0.0], algo.flush
    def timed_flush():
        a = time.perf_counter(); orig_flush(); t_flush[0] += time.perf_counter() - a
    algo.flush = timed_flush
    for batch in algo.batches(n):
        loss = algo.update(batch, learn=True); algo.step()
        log.append(loss)
    algo.flush()
    algo.flush = orig_flush
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
loop_rate = n / dt
print(f"   host time of the loop {t_host / n * 1e6:.1f} us/step, of which inside flush() {t_flush[0] / n * 1e6:.1f} us/step")
print(f"planned-batch update() loop, {dtype}:                    {loop_rate:8.0f} steps/s, {dt / n * 1e6:7.1f} us/step   (last value loss {float(log[-1]['value']):.4f})")
# ... and the reference's loop VERBATIM over env.train_dataloader, driven by the algo (attach_env(..., drive_loader=True))
env.train_dataloader.planner = algo
done, t0 = 0, None
for epoch in range(100):
    for batch in env.train_dataloader:
        loss = algo.update(batch, learn=True); algo.step()
        done += 1
        if done == 600:
            algo.flush(); torch.cuda.synchronize(); t0 = time.perf_counter()
        if done == 600 + n:
            break
    if done == 600 + n:
        break
algo.flush()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"for batch in env.train_dataloader (driven), {dtype}:       {n / dt:8.0f} steps/s, {dt / n * 1e6:7.1f} us/step")
env.train_dataloader.planner = None
algo.run(600)
torch.cuda.synchronize()
t0 = time.perf_counter()
algo.run(n)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"algo.run({n}), {dtype}:                                 {n / dt:8.0f} steps/s, {dt / n * 1e6:7.1f} us/step   loop / run = {loop_rate / (n / dt):.3f}")
