#!/usr/bin/env python3
"""x3_fwd_probe.py -- what bounds the split-bf16 forward GEMM?  The grouped layer-1 launch of a step (4 networks x [2048, 1418] x
[1418, 256]) as ONE problem of the same tile count ([8192, 1536] x [1536, 256] -> 256 tiles of 64 x 128), timed with the operands
placed differently: as they are; every A row aliased to row 0 (lda = 0: the activations' far-memory traffic is gone, the bytes
moved into LDS are not); every W row aliased (ldb = 0); both.  If the aliased runs are much faster the launch waits for far memory
(latency x bytes in flight), not for the LDS-DMA issue rate."""
import ctypes as C
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recnn_amd import _lib as L

L.load()
dev = torch.device("cuda")
M, N, K = int(os.environ.get("M", 8192)), 256, 1536
X = torch.randn(M, 2 * K, device=dev).to(torch.bfloat16)
W = (torch.randn(N, 2 * K, device=dev) * 0.05).to(torch.bfloat16)
out = torch.zeros(M, 2 * N, dtype=torch.bfloat16, device=dev)


def run(lda, ldb, reps=50):
    a = L.GemmArgs()
    C.memset(C.byref(a), 0, C.sizeof(a))
    a.dtype, a.M, a.N = L.BF16X3, M, N
    a.A[0], a.B[0], a.lda[0], a.ldb[0], a.K[0] = X.data_ptr(), W.data_ptr(), lda, ldb, 2 * K
    a.C, a.ldc, a.c_f32, a.relu = out.data_ptr(), 2 * N, 0, 1
    a.dx_scale, a.dw_splits = 1.0, 1
    for _ in range(5):
        L.call("recnn_gemm_fwd", C.byref(a), L.current_stream())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        L.call("recnn_gemm_fwd", C.byref(a), L.current_stream())
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


VARIANTS = [int(v) for v in os.environ.get("VARIANTS", "0").split(",")]
res = {}
PROBES = [int(v) for v in os.environ.get("PROBES", "0").split(",")]
for var, probe in [(v, p) for v in VARIANTS for p in PROBES]:
    L.load().recnn_debug_x3_fwd(var)
    L.load().recnn_debug_x3_ws_probe(probe)
    for name, lda, ldb in ((f"as is p{probe}", 2 * K, 2 * K), (f"both aliased p{probe}", 0, 0)):
        us = run(lda, ldb)
        big = (M // 64) * (N // 128) >= (48 if var >= 20 else 192)
        v = var % 20
        if v in (7, 10): tiles, rows_per_tile = -(-M // 128) * (N // 128), 256
        elif v == 8: tiles, rows_per_tile = -(-M // 128) * (N // 256), 384
        elif v == 9: tiles, rows_per_tile = (M // 64) * (N // 128), 192
        else: tiles, rows_per_tile = ((M // 64) * (N // 128), 192) if big else ((M // 32) * (N // 64), 96)
        gb = tiles * (2 * K // 128) * rows_per_tile * 256 / 1e9
        cus = min(tiles, 256)
        bclk = gb * 1e9 / cus / (us * 1e-6 * 2.4e9)
        print(f"variant {var:2d} M={M:5d} {name:18s} {us:7.2f} us   {tiles} tiles, {gb / (us * 1e-6) / 1e3:.1f} TB/s of LDS-DMA bytes, "
              f"{bclk:.1f} B/clk/CU at 2.4 GHz, {2.0 * M * N * K * 3 / (us * 1e-6) / 1e12:.0f} TFLOP/s executed")
        res[f"v{var}_{name.replace(' ', '_')}"] = {"us": us, "tiles": tiles, "dma_bytes": gb * 1e9, "B_per_clk_per_CU": bclk}
if os.environ.get("TRACE"):
    import numpy as np
    lib = L.load()
    lib.recnn_debug_x3_fwd(2)
    lib.recnn_debug_x3_ws_probe(0)
    RELU_MASK = int(os.environ.get("HASH", "1"))
    tr = torch.zeros(4096, 8, dtype=torch.int64, device=dev)
    bias = torch.randn(N, device=dev)
    a = L.GemmArgs()
    C.memset(C.byref(a), 0, C.sizeof(a))
    a.dtype, a.M, a.N = L.BF16X3, M, N
    a.A[0], a.B[0], a.lda[0], a.ldb[0], a.K[0] = X.data_ptr(), W.data_ptr(), 2 * K, 2 * K, 2 * K
    a.C, a.ldc, a.c_f32, a.relu = out.data_ptr(), 2 * N, 0, 1
    a.bias = bias.data_ptr()
    if RELU_MASK:
        a.mask_mode, a.seed, a.stream_id = L.MASK_HASH, 1234, 3
    a.dx_scale, a.dw_splits = 1.0, 1
    L.call("recnn_gemm_fwd", C.byref(a), L.current_stream())
    lib.recnn_debug_ws_trace(tr.data_ptr())
    L.call("recnn_gemm_fwd", C.byref(a), L.current_stream())
    torch.cuda.synchronize()
    lib.recnn_debug_ws_trace(None)
    t = tr.cpu().numpy()[: (M // 64) * (N // 128)]
    d = (t[:, 1:6] - t[:, 0:5]) / 2400.0
    print("x3 epilogue phases, us at 2.4 GHz (median / max over workgroups): barrier, epilogue -> LDS image, barrier, copy-out issue, stores acknowledged")
    print(np.round(np.median(d, 0), 2), np.round(d.max(0), 2))
L.load().recnn_debug_x3_fwd(-1)
L.load().recnn_debug_x3_ws_probe(0)
out_path = os.environ.get("OUT")
if out_path:
    import json
    with open(out_path, "w") as f:
        json.dump({"M": M, "N": N, "K_logical": K, "results": res}, f, indent=1)
