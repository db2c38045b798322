"""In-kernel phase timelines (shader-clock stamps) of the four launches that make the learning critic's per-step chain in the cycle schedule:
l1_gemm_kernel, mlp_tail_kernel, dw_adam_kernel (round 6: the dW GEMMs with the optimizer in their epilogue).  Eager DDPG steps on the split
forward at 2048 rows; the trace pointers are armed around ONE step, the stamps are those of the LAST launch of each kernel in that step.
usage: python tools/chain_trace.py > profiles/r06_chain_trace.txt"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recnn_amd import _lib as L
from recnn_amd.nn.engine import StepEngine
from recnn_amd._tune import set_default_tuning

S, A, H, B = 1290, 128, 256, int(os.environ.get("ROWS", "2048"))
dev = torch.device("cuda", 0)
torch.manual_seed(0)


def mk(inp, out):
    return {"w1": torch.randn(H, inp) * 0.03, "b1": torch.randn(H) * 0.1, "w2": torch.randn(H, H) * 0.06, "b2": torch.randn(H) * 0.1,
            "w3": torch.randn(out, H) * 0.3, "b3": torch.randn(out) * 0.3}


actor, critic = mk(S, A), mk(S + A, 1)
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
t_tail = torch.zeros(4 * 64 * max(1, B // 2048), 16, dtype=torch.int64, device=dev)
t_l1 = torch.zeros(4 * 256 * max(1, B // 2048), 16, dtype=torch.int64, device=dev)
t_dwa = torch.zeros(2 * 256, 8, dtype=torch.int64, device=dev)
lib = L.load()
lib.recnn_debug_tail_trace(L.ptr(t_tail))
lib.recnn_debug_l1_trace(L.ptr(t_l1))
lib.recnn_debug_dwadam_trace(L.ptr(t_dwa))
eng.step(B, True, 1)
torch.cuda.synchronize()
lib.recnn_debug_tail_trace(None)
lib.recnn_debug_l1_trace(None)
lib.recnn_debug_dwadam_trace(None)


def show(name, tr, labels):
    tr = tr.cpu().numpy()
    rows = tr[tr[:, 0] > 0]
    print(f"{name}: {len(rows)} workgroups stamped (shader clocks from the workgroup's entry)")
    for k, lab in labels:
        good = rows[:, k] > 0
        if good.any():
            v = (rows[good, k] - rows[good, 0])
            print(f"   {lab:34s} min {v.min():7d} median {int(np.median(v)):7d} max {v.max():7d}   ({good.sum()} workgroups)")
    if len(rows):
        print(f"   first entry -> last stamp of the launch: {int(rows.max() - rows[:, 0].min())} clocks")


show("l1_gemm_kernel (learning critic, 64 x 64 tiles)", t_l1, ((1, "prologue issued + bias"), (2, "k loop done"), (3, "end")))
show("mlp_tail_kernel (learning critic)", t_tail, ((1, "operands landed"), (2, "layer 2 multiplied"), (3, "h2 epilogue"), (4, "q dots"), (5, "head"),
                                                  (6, "dw3 sums"), (7, "u2"), (8, "U mfma (+db2 sums)"), (9, "U written"), (10, "end")))
show("dw_adam_kernel (tile workgroups)", t_dwa, ((1, "state requested (+ scalars, wave 7)"), (2, "first stage landed"), (3, "contraction done"),
                                                 (4, "partial tiles in LDS"), (5, "optimizer arithmetic done"), (6, "stores issued"),
                                                 (7, "stores acknowledged")))
prof = eng.profile(B, policy=False, n_steps=20)
print("eager launch times (HIP events, us):", ", ".join(f"{n} {ms * 1e3:.2f}" for n, ms, _ in prof))
eng.set_tuning(split_fwd=2, dw_fuse=0)
prof = eng.profile(B, policy=False, n_steps=20)
print("... with dw_fuse = 0:               ", ", ".join(f"{n} {ms * 1e3:.2f}" for n, ms, _ in prof))
