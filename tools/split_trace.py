"""Phase timelines of the split forward's kernels (csrc/l1gemm.hip, csrc/mlpt.hip) from in-kernel shader-clock stamps.
usage: python tools/split_trace.py   (eager DDPG step, 2048 rows: which launch a trace belongs to is selected by arming the
trace pointer around ONE eager step and reading the stamps of the LAST launch of each kernel = the learning critic's)"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recnn_amd import _lib as L
from recnn_amd.nn.engine import StepEngine

S, A, H, B = 1290, 128, 256, 2048
dev = torch.device("cuda", 0)
torch.manual_seed(0)

def mk(inp, out):
    return {"w1": torch.randn(H, inp) * 0.03, "b1": torch.randn(H) * 0.1, "w2": torch.randn(H, H) * 0.06, "b2": torch.randn(H) * 0.1,
            "w3": torch.randn(out, H) * 0.3, "b3": torch.randn(out) * 0.3}
actor, critic = mk(S, A), mk(S + A, 1)
from recnn_amd._tune import set_default_tuning
set_default_tuning(split_fwd=2)
eng = StepEngine("ddpg", S, A, H, B, dtype="bf16", mask_mode="hash", seed=1, device=dev)
for ni, p in ((L.NET_POLICY, actor), (L.NET_TARGET_POLICY, actor), (L.NET_VALUE1, critic), (L.NET_TARGET_VALUE1, critic)):
    eng.load_params(ni, p)
eng.set_hyper(policy_opt=dict(lr=1e-5), value_opt=dict(lr=1e-5))
eng.set_counters()
eng.pack_batch(torch.randn(B, S), torch.randn(B, A), torch.randn(B), torch.randn(B, S), (torch.rand(B) < 0.1).float())
for t in range(5):
    eng.step(B, True, 1)
torch.cuda.synchronize()
t_tail = torch.zeros(4 * 64, 16, dtype=torch.int64, device=dev)
t_l1 = torch.zeros(4 * 256, 16, dtype=torch.int64, device=dev)
L.load().recnn_debug_tail_trace(L.ptr(t_tail))
L.load().recnn_debug_l1_trace(L.ptr(t_l1))
eng.step(B, True, 1)
torch.cuda.synchronize()
L.load().recnn_debug_tail_trace(None)
L.load().recnn_debug_l1_trace(None)

def show(name, tr, labels):
    tr = tr.cpu().numpy()
    ok = tr[:, 0] > 0
    rows = tr[ok]
    print(f"{name}: {len(rows)} workgroups (the step's last launch of this kernel overwrote the earlier ones)")
    for k, lab in labels:
        good = rows[:, k] > 0
        if good.any():
            v = (rows[good, k] - rows[good, 0])
            print(f"   {lab:28s} min {v.min():7d} median {int(np.median(v)):7d} max {v.max():7d}")

show("l1_gemm (learning critic, 64 x 64 tiles)", t_l1, ((1, "prologue issued + bias"), (2, "k loop done"), (3, "end")))
show("mlp_tail (learning critic)", t_tail, ((1, "operands landed"), (2, "layer 2 multiplied"), (3, "h2 epilogue"), (4, "q dots"), (5, "head"), (6, "dw3 sums"),
                                          (7, "u2"), (8, "U mfma (+db2 sums)"), (9, "U written"), (10, "end")))
