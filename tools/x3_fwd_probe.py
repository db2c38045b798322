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


for name, lda, ldb in (("as is", 2 * K, 2 * K), ("A rows aliased", 0, 2 * K), ("W rows aliased", 2 * K, 0), ("both aliased", 0, 0)):
    us = run(lda, ldb)
    gb = (M // 64) * (N // 128) * (2 * K // 128) * 48 * 1024 / 1e9
    print(f"{name:16s} {us:7.2f} us   ({gb / (us * 1e-6) / 1e3:.1f} TB/s of LDS-DMA bytes, {2.0 * M * N * K * 3 / (us * 1e-6) / 1e12:.0f} TFLOP/s executed)")
