"""In-kernel phase timeline (shader-clock stamps by thread 0 of every workgroup) of mlp_frozen_kernel in the cycle schedule's run graphs
(the trace pointer is captured into the graphs: armed BEFORE attach_env builds them).  usage: [DRIVER=1] python tools/frozen_trace.py"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import recnn_amd
from recnn_amd import _lib as L
from recnn_amd.nn import fused

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
items, ratings, off, lens = bench.synthetic_store(0)
table = torch.randn(bench.N_ITEMS, bench.EMB, generator=torch.Generator().manual_seed(0))
env = recnn_amd.data.env.FrameEnv.from_store(table, items, ratings, off, frame_size=bench.FRAME, batch_size=25, device=dev, test_fraction=0.0)
fused.set_defaults(dtype="bf16", mask_mode="hash", seed=1234)
recnn_amd.nn.algo.set_default_optimizer("adam")
torch.manual_seed(0)
algo = recnn_amd.nn.DDPG(recnn_amd.nn.Actor(bench.STATE, bench.EMB, bench.HIDDEN, 6e-1), recnn_amd.nn.Critic(bench.STATE, bench.EMB, bench.HIDDEN, 54e-2)).to(dev)
tr = torch.zeros(2 * 512, 16, dtype=torch.int64, device=dev)
lib = L.load()
lib.recnn_debug_frozen_trace(L.ptr(tr))
algo.attach_env(env, rows_per_batch=2048, users_per_batch=None, shard=(0, 1))
stream = torch.cuda.Stream(device=dev)
with torch.cuda.stream(stream):
    if os.environ.get("DRIVER"):     # the driver's request: 5 warm-up steps, then 20 = segments of 6 + 10 + 4 (the short ones on 64-row workgroups)
        algo.run(5)
        algo.prepare_run(20, first_step=5)
        algo.run(20)
    else:
        algo.run(200)
    torch.cuda.synchronize()
lib.recnn_debug_frozen_trace(None)
t = tr.cpu().numpy()
rows = t[t[:, 0] > 0]
labels = [(1, "slab 0 landed"), (2, "8 slabs multiplied"), (3, "16 slabs multiplied"), (4, "layer 1 done"), (5, "h1 epilogue"), (6, "layer 2 done"),
          (7, "h2 epilogue"), (8, "layer 3 multiplied (actor)"), (9, "end")]
print(f"{len(rows)} workgroups stamped (every launch overwrites its workgroups' rows: the cycle's second launch, and the first one's rows it does not cover); shader clocks from the workgroup's entry")
for k, lab in labels:
    good = rows[:, k] > 0
    if good.any():
        v = rows[good, k] - rows[good, 0]
        print(f"   {lab:30s} min {v.min():7d} median {int(np.median(v)):7d} max {v.max():7d}   ({good.sum()} workgroups)")
