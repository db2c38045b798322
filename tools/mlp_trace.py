"""Phase timeline of the fused row-panel MLP forward (csrc/mlps.hip) from in-kernel shader-clock stamps.
usage: python tools/mlp_trace.py [probe_bits]   (1 = no MFMA work, 2 = no DMA)"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recnn_amd import _lib as L
from recnn_amd.nn.engine import StepEngine

probe = int(sys.argv[1]) if len(sys.argv) > 1 else 0
kernel = 3   # csrc/mlps.hip, the only fused forward left
S, A, H, B = 1290, 128, 256, 2048
dev = torch.device("cuda", 0)
torch.manual_seed(0)

def mk(inp, out):
    return {"w1": torch.randn(H, inp) * 0.03, "b1": torch.randn(H) * 0.1, "w2": torch.randn(H, H) * 0.06, "b2": torch.randn(H) * 0.1,
            "w3": torch.randn(out, H) * 0.3, "b3": torch.randn(out) * 0.3}
actor, critic = mk(S, A), mk(S + A, 1)
eng = StepEngine("ddpg", S, A, H, B, dtype="bf16", mask_mode="hash", seed=1, device=dev)
for ni, p in ((L.NET_POLICY, actor), (L.NET_TARGET_POLICY, actor), (L.NET_VALUE1, critic), (L.NET_TARGET_VALUE1, critic)):
    eng.load_params(ni, p)
eng.set_hyper(policy_opt=dict(lr=1e-5), value_opt=dict(lr=1e-5))
eng.set_counters()
eng.pack_batch(torch.randn(B, S), torch.randn(B, A), torch.randn(B), torch.randn(B, S), (torch.rand(B) < 0.1).float())
trace = torch.zeros(1024, 32, dtype=torch.int64, device=dev)
L.load().recnn_debug_mlp_probe(probe)
for t in range(5):
    eng.step(B, True, 1)
torch.cuda.synchronize()
L.load().recnn_debug_mlp_trace(L.ptr(trace))
eng.step(B, True, 1)
torch.cuda.synchronize()
L.load().recnn_debug_mlp_trace(None)
tr = trace.cpu().numpy()
npanel = B // 32
names = ["tc_producer", "critic", "target_actor+tail+head", "actor"]
t_start = tr[:npanel * 4, 0][tr[:npanel * 4, 0] > 0].min()
t_end = tr[:npanel * 4, 9].max()
print(f"kernel {kernel} probe {probe}: launch span {t_end - t_start} ticks (first workgroup start -> last workgroup end)")
labels = {13: "L2 slab 2 before wait", 14: "L2 slab 2 after wait+barrier", 15: "L2 mma done (before barrier)", 1: "setup done", 10: "L1 slab 2", 11: "L1 slab 12 (before wait)", 12: "L1 slab 12 (after wait+barrier)", 2: "L1 done", 3: "epilogue 1 done",
          4: "L2 done", 5: "epilogue 2 done", 6: "L3 done / critic head done", 7: "tails done", 9: "end"}
for pi, nm in enumerate(names):
    rows = tr[pi * npanel:(pi + 1) * npanel]
    print(f"== {nm}: start offset vs launch start: median {np.median(rows[:, 0] - t_start):.0f}, max {np.max(rows[:, 0] - t_start):.0f} ticks")
    if kernel == 3:
        seq = [(1, "setup done"), (2, "L1 done"), (3, "epilogue 1 done"), (10, "L2 slab 0 landed+issued"), (24, "  h1 panel stored (issued)"), (25, "  slab 0 multiplied"), (22, "  slab 1: vmcnt wait over"),
               (23, "  slab 1: barrier passed"), (11, "L2 slab 1 (next slab issued)"), (12, "L2 slab 2"),
               (13, "L2 slab 3"), (4, "L2 mma done"), (5, "epilogue 2 done"), (14, "L3 slab 0 landed+issued"), (15, "L3 slab 1"), (16, "L3 mma done"),
               (6, "outputs stored / critic q done"), (17, "tail: part slab landed"), (18, "tail: action slabs multiplied"),
               (19, "tail: epilogue 1"), (20, "tail: W2 multiplied"), (21, "tail: epilogue 2"), (7, "tails done (q dots)"), (9, "end")]
        prev = None
        for k, lab in seq:
            v = rows[:, k]
            if (v > 0).all():
                t = np.median(v - rows[:, 0])
                print(f"   {lab:34s} {t:9.0f}   (+{t - prev:6.0f})" if prev is not None else f"   {lab:34s} {t:9.0f}")
                prev = t
        continue
    for k in ((1, 2, 3, 4, 5, 6, 7, 9) if kernel == 3 else (1, 10, 11, 12, 2, 3, 13, 14, 4, 15, 5, 6, 7, 9, 16 + 2, 16 + 3, 16 + 13, 16 + 14, 16 + 4, 16 + 15, 16 + 5, 16 + 6, 16 + 9)):
        v = rows[:, k]
        if (v > 0).all():
            print(f"   {('w15 ' if k >= 16 else '') + labels[k % 16 if k >= 16 else k]:34s} median {np.median(v - rows[:, 0]):9.0f}  max {np.max(v - rows[:, 0]):9.0f} ticks since workgroup start")
    if kernel == 3 and (rows[:, 15] > 0).all():
        print("   L2 slab 1: enter %.0f -> waited %.0f -> barrier %.0f -> issued %.0f -> mma done %.0f -> slab 2 mma done %.0f (ticks since workgroup start)"
              % tuple(np.median(rows[:, k] - rows[:, 0]) for k in (10, 11, 12, 13, 14, 15)))
    elif (rows[:, 11] > 0).all():
        nsl = 5 if kernel == 0 else 10
        print(f"   per L1 slab: {np.median((rows[:, 11] - rows[:, 10]) / nsl):.0f} ticks; wait+barrier of one slab: {np.median(rows[:, 12] - rows[:, 11]):.0f}")
