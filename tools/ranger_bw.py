"""ranger_bw.py -- the flat Ranger pass over a catalogue-sized weight ([2048, 101290] fp32 + its bf16 row-padded copy, REINFORCE's critic
layer 1): achieved HBM bandwidth with 16-byte lanes (every array aligned) against 4-byte lanes (the parameter at a 4-byte offset).
usage: python tools/ranger_bw.py [rows cols]"""
import ctypes as C
import json
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recnn_amd import _lib as L

L.load()
dev = torch.device("cuda")
rows, cols = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2048, 101290)
n = rows * cols
ld = (cols + 63) // 64 * 64
out = {"rows": rows, "cols": cols, "bytes_per_element": 30, "results": {}}
for name, off in (("16-byte lanes", 0), ("4-byte lanes", 1)):
    base = torch.randn(n + 4, device=dev)
    p = base[off:off + n]
    g = torch.randn(n, device=dev) * 1e-2
    m, v, slow = torch.zeros(n, device=dev), torch.zeros(n, device=dev), p.clone()
    shadow = torch.zeros(rows, ld, dtype=torch.bfloat16, device=dev)
    sh = L.ShadowOut()
    sh.dst, sh.cols, sh.ld, sh.bf16 = shadow.data_ptr(), cols, ld, 1
    def step(t):
        L.call("recnn_ranger_flat_shadow", L.ptr(p), L.ptr(g), L.ptr(m), L.ptr(v), L.ptr(slow), n, 1e-3, 0.95, 0.999, 1e-5, 1e-2, 0.5, 6, 5.0,
               t, 1.0, C.byref(sh), L.current_stream())
    for t in range(1, 4):
        step(t)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = [7, 8, 9, 10, 11, 13, 14, 15, 16, 17]          # (no Lookahead sync step among them)
    e0.record()
    for t in reps:
        step(t)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / len(reps) * 1e3
    out["results"][name] = {"us": round(us, 1), "TBs": round(n * 30 / (us * 1e-6) / 1e12, 2)}
    print(name, round(us, 1), "us", round(n * 30 / (us * 1e-6) / 1e12, 2), "TB/s (30 B per element: p, g, m, v read; p, m, v, bf16 copy written)")
    assert torch.equal(shadow[:, :cols], p.view(rows, cols).to(torch.bfloat16))
    del base, p, g, m, v, slow, shadow
if os.environ.get("OUT"):
    json.dump(out, open(os.environ["OUT"], "w"), indent=1)
