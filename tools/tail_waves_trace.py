"""Per-WAVE phase stamps of mlp_tail_kernel's learning-critic workgroups (bit 0 of the trace pointer): how far apart the 16 waves of a
workgroup reach each stamp, i.e. what the barriers absorb.  usage: python tools/tail_waves_trace.py"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recnn_amd import _lib as L
from recnn_amd.nn.engine import StepEngine
from recnn_amd._tune import set_default_tuning

S, A, H, B = 1290, 128, 256, 2048
dev = torch.device("cuda", 0)
torch.manual_seed(0)
mk = lambda inp, out: {"w1": torch.randn(H, inp) * 0.03, "b1": torch.randn(H) * 0.1, "w2": torch.randn(H, H) * 0.06, "b2": torch.randn(H) * 0.1,
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
t_tail = torch.zeros(4 * 256, 16, 16, dtype=torch.int64, device=dev)
import ctypes as C
L.load().recnn_debug_tail_trace(C.c_void_p(t_tail.data_ptr() | 1))
eng.step(B, True, 1)
torch.cuda.synchronize()
L.load().recnn_debug_tail_trace(None)
tr = t_tail.cpu().numpy()
wgs = [w for w in range(tr.shape[0]) if tr[w, 0, 10] > 0 and tr[w, 0, 9] > 0]       # learning-critic workgroups (they reach stamp 9)
print(f"{len(wgs)} learning-critic workgroups; per stamp: median over workgroups of (earliest wave, latest wave) relative to the workgroup's first entry")
labels = {1: "operands landed", 2: "layer 2 multiplied", 3: "h2 epilogue", 4: "q dots / burst", 5: "B2 passed", 6: "dz2 + column sums", 7: "B3 passed",
          8: "U mfma", 9: "U written", 10: "end"}
for k, lab in labels.items():
    lo, hi = [], []
    for w in wgs:
        t0 = tr[w, :, 0][tr[w, :, 0] > 0].min()
        v = tr[w, :, k]
        v = v[v > 0]
        if len(v):
            lo.append(v.min() - t0); hi.append(v.max() - t0)
    if lo:
        print(f"   {lab:22s} earliest wave {int(np.median(lo)):6d}   latest wave {int(np.median(hi)):6d}   skew {int(np.median(hi)) - int(np.median(lo)):5d}")
