#!/usr/bin/env python3
"""What a standalone launch of the gather's size can reach on this box: HIP-event time of (a) a device-to-device copy moving
the gather's algorithmic bytes (22.9 MB per launch at 2048 rows with bf16 rows: read + write), (b) the same at 10x the
bytes, (c) a trivially small copy (launch + event floor).  Prints one JSON line.  Context for `roofline_gather.frac`."""
import json
import torch

dev = torch.device("cuda:0")


def timed(n_bytes, iters=300):
    src = torch.empty(n_bytes // 2, dtype=torch.uint8, device=dev).random_(0, 255)
    dst = torch.empty_like(src)
    for _ in range(20):
        dst.copy_(src)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in ev:
        a.record()
        dst.copy_(src)
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2] * 1e3      # median, us


out = {}
for name, nb in (("gather_bytes", 22_921_216), ("ten_times", 229_212_160), ("tiny", 4096)):
    us = timed(nb)
    out[name] = {"bytes_moved": nb, "us": us, "GBps": nb / us / 1e3, "frac_of_8TBps": nb / us / 1e3 / 8000.0}
print(json.dumps(out))
