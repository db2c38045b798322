#!/usr/bin/env python3
"""Micro-benchmark of the fused scoring + top-K kernel: B=2048 generated actions against the
ML20M-sized (26,744) and the 100k-item catalog, k=10.  Prints one JSON line per case with the fp32-MFMA roofline
fraction and a numpy (all host cores through BLAS) baseline on a bounded sample."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import retrieval_oracle as R  # noqa: E402
from recnn_amd.retrieval import FlatIndex  # noqa: E402

dev = torch.device("cuda")
rng = np.random.default_rng(0)
for N in (26744, 100000):
    B, k = 2048, 10
    table = rng.standard_normal((N, 128)).astype(np.float32)
    q = rng.standard_normal((B, 128)).astype(np.float32)
    idx = FlatIndex(torch.from_numpy(table).to(dev), "L2")
    qd = torch.from_numpy(q).to(dev)
    for _ in range(3):
        idx.search(qd, k)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    reps = 20
    for _ in range(reps):
        idx.search(qd, k)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    flops = 2.0 * B * N * 128
    t0 = time.perf_counter()
    R.topk(q[:256], table, "L2", k)
    cpu_s = (time.perf_counter() - t0) * (B / 256)
    print(json.dumps({"metric": "top-K item search, queries/s", "value": B / (ms * 1e-3), "ms": ms, "B": B, "N": N, "k": k,
                      "roofline": {"bound": "mfma", "achieved": flops / (ms * 1e-3) / 1e12, "peak": 157.3, "unit": "TFLOP/s",
                                   "frac": flops / (ms * 1e-3) / 1e12 / 157.3},
                      "cpu_baseline": {"value": B / cpu_s, "unit": "queries/s", "kind": "port",
                                       "sample": "256 queries, float64 numpy oracle, scaled"}}))
