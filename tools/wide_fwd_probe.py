#!/usr/bin/env python3
"""wide_fwd_probe.py -- the catalogue-wide bf16 forward product of REINFORCE ([256, K] x [K, 100k], recnn/nn/models.py:93-95) timed with
the weight rows at different PITCHES: 4096-byte rows put the same 128-byte piece of every tile row -- and of every workgroup, whose
panels start 512 KB apart -- on the same few HBM channels; a pitch that is not a power of two spreads them.
usage: python tools/wide_fwd_probe.py   (env: M, N, K, PADS="0,64,128", OUT=json path)"""
import ctypes as C
import json
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recnn_amd import _lib as L

lib = L.load()
dev = torch.device("cuda")
M, N, K = int(os.environ.get("M", 256)), int(os.environ.get("N", 100000)), int(os.environ.get("K", 2048))
PROBES = [int(v) for v in os.environ.get("PROBES", "0").split(",")]   # recnn_debug_x3_ws_probe bits (the wave-specialised tiles only)
PADS = [int(v) for v in os.environ.get("PADS", "0,64,128,192").split(",")]
X = torch.randn(M, K, device=dev).to(torch.bfloat16)
res = {}
for tall in [int(v) for v in os.environ.get("TILES", "1,0").split(",")]:
    lib.recnn_debug_wide_ws(tall)
    for pad, probe in [(pd, pr) for pd in PADS for pr in PROBES]:
        lib.recnn_debug_x3_ws_probe(probe)
        ld = K + pad
        Wp = torch.zeros(N, ld, dtype=torch.bfloat16, device=dev)
        Wp[:, :K] = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
        CF32 = int(os.environ.get("C_F32", "1"))
        out = torch.zeros(M, N, device=dev, dtype=torch.float32 if CF32 else torch.bfloat16)
        a = L.GemmArgs()
        C.memset(C.byref(a), 0, C.sizeof(a))
        a.dtype, a.M, a.N = L.BF16, M, N
        a.A[0], a.B[0], a.lda[0], a.ldb[0], a.K[0] = X.data_ptr(), Wp.data_ptr(), K, ld, K
        a.C, a.ldc, a.c_f32, a.relu = out.data_ptr(), N, CF32, 0
        a.dx_scale, a.dw_splits = 1.0, 1
        for _ in range(3):
            L.call("recnn_gemm_fwd", C.byref(a), L.current_stream())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            L.call("recnn_gemm_fwd", C.byref(a), L.current_stream())
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        name = "%s, weight pitch %d B" % ({0: "128 x 128 (round 3: every wave loads and multiplies, 2 x 64 KB stages)", 1: "256 x 128, 3 stages", 2: "128 x 128, 4 stages",
                                            3: "64 x 256, 4 stages", 4: "128 x 256, 3 stages", 5: "64 x 128, 6 stages"}[tall], ld * 2) + (" probe %d" % probe if probe else "")
        res[name] = {"us": round(us, 1), "TFLOPs": round(2.0 * M * N * K / (us * 1e-6) / 1e12), "weight_GBs": round(N * K * 2 / (us * 1e-6) / 1e9)}
        print(name, res[name])
        del Wp, out
lib.recnn_debug_wide_ws(1)
lib.recnn_debug_x3_ws_probe(0)
if os.environ.get("TRACE"):
    tr = torch.zeros(4096, 8, dtype=torch.int64, device=dev)
    Wp = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
    out = torch.zeros(M, N, device=dev)
    a = L.GemmArgs()
    C.memset(C.byref(a), 0, C.sizeof(a))
    a.dtype, a.M, a.N = L.BF16, M, N
    a.A[0], a.B[0], a.lda[0], a.ldb[0], a.K[0] = X.data_ptr(), Wp.data_ptr(), K, K, K
    a.C, a.ldc, a.c_f32, a.relu = out.data_ptr(), N, 1, 0
    a.dx_scale, a.dw_splits = 1.0, 1
    L.call("recnn_gemm_fwd", C.byref(a), L.current_stream())
    lib.recnn_debug_ws_trace(tr.data_ptr())
    L.call("recnn_gemm_fwd", C.byref(a), L.current_stream())
    torch.cuda.synchronize()
    lib.recnn_debug_ws_trace(None)
    t = tr.cpu().numpy()[: -(-N // 128)]
    d = (t[:, 1:6] - t[:, 0:5]) / 2400.0     # s_memtime counts shader clocks (~2.4 GHz) -> us
    import numpy as np
    print("epilogue phases, us (median / max over workgroups): barrier, math -> LDS, barrier, copy-out issue, stores acknowledged")
    print(np.round(np.median(d, 0), 2), np.round(d.max(0), 2))
if os.environ.get("OUT"):
    json.dump({"M": M, "N": N, "K": K, "results": res}, open(os.environ["OUT"], "w"), indent=1)
