#!/usr/bin/env python3
"""catalogue_gemm_probe.py -- the catalogue-wide policy-head product [rows, hidden] x [hidden, n_items] (recnn/nn/models.py:93-95 at a
100k-item catalogue) through recnn_gemm_fwd: time per launch (HIP events on the launch stream), fraction of the
dense MFMA peak of the compute type, max-norm error against torch."""
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recnn_amd import _lib as L

L.load()
dev = torch.device("cuda")
M, K, N = int(os.environ.get("ROWS", 256)), 2048, int(os.environ.get("ITEMS", 100000))
PEAK = {"bf16": 2500.0, "fp32": 157.3}
out_json = {}
for dtype in ("bf16", "fp32"):
    tdt = torch.bfloat16 if dtype == "bf16" else torch.float32
    x = torch.randn(M, K, device=dev).to(tdt)
    w = (torch.randn(N, K, device=dev) * 0.02).to(tdt)
    b = torch.randn(N, device=dev)
    ldn = (N + 63) // 64 * 64
    out = torch.zeros(M, ldn, device=dev)
    a = L.GemmArgs()
    C.memset(C.byref(a), 0, C.sizeof(a))
    a.dtype, a.M, a.N = L.DTYPES[dtype], M, N
    a.A[0], a.B[0], a.lda[0], a.ldb[0], a.K[0] = x.data_ptr(), w.data_ptr(), K, K, K
    a.C, a.ldc, a.c_f32, a.bias = out.data_ptr(), ldn, 1, b.data_ptr()
    a.dx_scale, a.dw_splits = 1.0, 1
    for _ in range(3):
        L.call("recnn_gemm_fwd", C.byref(a), L.current_stream())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        L.call("recnn_gemm_fwd", C.byref(a), L.current_stream())
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    ref = x.float() @ w.float().t() + b
    err = float((out[:, :N] - ref).abs().max() / ref.abs().max())
    tf = 2.0 * M * N * K / (us * 1e-6) / 1e12
    out_json[dtype] = {"us": round(us, 1), "tflops": round(tf, 1), "frac_of_peak": round(tf / PEAK[dtype], 3), "max_rel_err": err}
print(json.dumps({"shape": [M, K, N], "launch": out_json}))
