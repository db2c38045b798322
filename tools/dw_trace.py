"""Phase timeline of dw_opt_kernel (csrc/dwopt.hip) from in-kernel shader-clock stamps.  usage: python tools/dw_trace.py [probe_bits] [fuse_mode]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recnn_amd import _lib as L
from recnn_amd.nn.engine import StepEngine

probe = int(sys.argv[1]) if len(sys.argv) > 1 else 0
fuse = int(sys.argv[2]) if len(sys.argv) > 2 else 1
S, A, H, B = 1290, 128, 256, 2048
dev = torch.device("cuda", 0)
torch.manual_seed(0)

def mk(inp, out):
    return {"w1": torch.randn(H, inp) * 0.03, "b1": torch.randn(H) * 0.1, "w2": torch.randn(H, H) * 0.06, "b2": torch.randn(H) * 0.1,
            "w3": torch.randn(out, H) * 0.3, "b3": torch.randn(out) * 0.3}
actor, critic = mk(S, A), mk(S + A, 1)
L.load().recnn_tune_dw_fuse(fuse)
eng = StepEngine("ddpg", S, A, H, B, dtype="bf16", mask_mode="hash", seed=1, device=dev)
for ni, p in ((L.NET_POLICY, actor), (L.NET_TARGET_POLICY, actor), (L.NET_VALUE1, critic), (L.NET_TARGET_VALUE1, critic)):
    eng.load_params(ni, p)
eng.set_hyper(policy_opt=dict(lr=1e-5), value_opt=dict(lr=1e-5))
eng.set_counters()
eng.pack_batch(torch.randn(B, S), torch.randn(B, A), torch.randn(B), torch.randn(B, S), (torch.rand(B) < 0.1).float())
GX = 92
trace = torch.zeros(3 * GX, 8, dtype=torch.int64, device=dev)
L.load().recnn_tune_dw_probe(probe)
for t in range(5):
    eng.step(B, True, 1)
torch.cuda.synchronize()
L.load().recnn_tune_dw_trace(L.ptr(trace))
eng.step(B, True, 1)
torch.cuda.synchronize()
L.load().recnn_tune_dw_trace(None)
tr = trace.cpu().numpy().reshape(3, GX, 8)
t0 = tr[:, :, 0][tr[:, :, 0] > 0].min()
print(f"probe {probe} fuse {fuse}: stamps relative to the first workgroup's start (shader clock ticks, ~100 MHz s_memtime => x10 ns? see below)")
for y, name, n in ((0, "vec", 25), (1, "W2 tiles", 16), (2, "W1 tiles", 92)):
    rows = tr[y, :n].astype(np.int64)
    rel = np.where(rows > 0, rows - t0, -1)
    labs = ((1, "staged + scalars"), (2, "row loop done"), (7, "end")) if y == 0 else ((1, "k loop done"), (2, "barrier"), (3, "tile in LDS"), (7, "end"))
    for k, lab in labs:          # per-workgroup durations since its own start (clocks of different XCDs are not comparable)
        ok = (rows[:, k] > 0) & (rows[:, 0] > 0)
        v = (rows[:, k] - rows[:, 0])[ok]
        if len(v):
            print(f"  {name:9s} {lab:22s} min {v.min():8d} median {int(np.median(v)):8d} max {v.max():8d}")
print("empty workgroups of y=0:", int((tr[0, 25:, 7] > 0).sum()), " span of all stamps:", int(tr.max() - t0))
