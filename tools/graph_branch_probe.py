#!/usr/bin/env python3
"""graph_branch_probe.py -- do two INDEPENDENT kernel chains inside one hipGraph overlap on this runtime?  (Round 3 measured that a
small side branch cost more than it hid; this asks the question for whole chains of latency-bound GEMM launches.)
Chain = n launches of the split-bf16 forward GEMM [M, 1536] x [1536, 256] (each launch depends on the previous one through the
stream).  Timed as graph replays: one chain alone, two chains back to back on one stream, two chains on forked streams."""
import ctypes as C
import json
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recnn_amd import _lib as L

L.load()
dev = torch.device("cuda")
N, K = 256, 1536


def mk(M):
    X = torch.randn(M, 2 * K, device=dev).to(torch.bfloat16)
    W = (torch.randn(N, 2 * K, device=dev) * 0.05).to(torch.bfloat16)
    out = torch.zeros(M, 2 * N, dtype=torch.bfloat16, device=dev)
    a = L.GemmArgs()
    C.memset(C.byref(a), 0, C.sizeof(a))
    a.dtype, a.M, a.N = L.BF16X3, M, N
    a.A[0], a.B[0], a.lda[0], a.ldb[0], a.K[0] = X.data_ptr(), W.data_ptr(), 2 * K, 2 * K, 2 * K
    a.C, a.ldc, a.c_f32, a.relu = out.data_ptr(), 2 * N, 0, 1
    a.dx_scale, a.dw_splits = 1.0, 1
    return a, (X, W, out)


def chain(a, n, stream):
    for _ in range(n):
        L.call("recnn_gemm_fwd", C.byref(a), C.c_void_p(stream.cuda_stream))


def time_graph(build, reps=30):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        build(s)            # warm-up (kernel attributes)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            build(s)
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(reps):
            g.replay()
        e1.record(s)
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


res = {}
for M in (2048, 512):
    a1, k1 = mk(M)
    a2, k2 = mk(M)
    n = 8
    side = torch.cuda.Stream()

    def one(s):
        chain(a1, n, s)

    def seq(s):
        chain(a1, n, s)
        chain(a2, n, s)

    def par(s):
        side.wait_stream(s)
        chain(a1, n, s)
        chain(a2, n, side)
        s.wait_stream(side)

    def inter(s):       # the two chains interleaved on ONE stream (what a single-stream schedule does)
        for _ in range(n):
            chain(a1, 1, s)
            chain(a2, 1, s)

    r = {"one_chain_us": time_graph(one), "two_sequential_us": time_graph(seq), "two_interleaved_us": time_graph(inter), "two_forked_us": time_graph(par)}
    res[f"M={M}"] = r
    print(f"M={M}: {n} launches/chain: one chain {r['one_chain_us']:.1f} us, two sequential {r['two_sequential_us']:.1f}, "
          f"interleaved {r['two_interleaved_us']:.1f}, two on forked streams {r['two_forked_us']:.1f}")
out = os.environ.get("OUT")
if out:
    json.dump(res, open(out, "w"), indent=1)
