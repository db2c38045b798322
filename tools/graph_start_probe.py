"""When does the GPU start a hipGraph?  A captured chain of N small kernels timed twice: launched on an idle stream (launch call -> done = start
delay + execution) and launched behind a spin kernel that keeps the GPU busy while the host enqueues (launch call -> done = spin + execution): the
difference is what the idle launch waits before its first kernel runs, next to the host time of hipGraphLaunch itself.
usage: python tools/graph_start_probe.py"""
import time
import numpy as np
import torch
dev = torch.device("cuda", 0)
flag_d = torch.zeros(1, dtype=torch.int32, device=dev)
flag_h = torch.zeros(1, dtype=torch.int32).pin_memory()
x = torch.zeros(1 << 16, device=dev)
s = torch.cuda.Stream(device=dev)
res = {}
with torch.cuda.stream(s):
    for n in (10, 50, 100, 200, 400):
        g = torch.cuda.CUDAGraph()
        flag_d.zero_()
        x.add_(1.0)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            flag_d.fill_(1)
            flag_h.copy_(flag_d, non_blocking=True)
            for _ in range(n):
                x.add_(1.0)
        # calibrate the spin kernel to ~1.5 ms
        if "cyc" not in res:
            torch.cuda.synchronize(); t0 = time.perf_counter(); torch.cuda._sleep(10_000_000); torch.cuda.synchronize()
            res["cyc"] = int(10_000_000 * 1.5e-3 / (time.perf_counter() - t0))
        rec = []
        for r in range(30):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            g.replay()
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            torch.cuda._sleep(res["cyc"])
            t3 = time.perf_counter()
            g.replay()
            t4 = time.perf_counter()
            torch.cuda.synchronize()
            t5 = time.perf_counter()
            torch.cuda._sleep(res["cyc"])
            torch.cuda.synchronize()
            t6 = time.perf_counter()
            rec.append(((t1 - t0) * 1e6, (t2 - t0) * 1e6, (t5 - t3) * 1e6, (t6 - t5) * 1e6))
        m = np.median(np.array(rec[5:]), axis=0)
        print(f"{n + 2:4d} nodes: hipGraphLaunch host time {m[0]:6.1f} us | idle stream: launch -> done {m[1]:7.1f} us | behind a {m[3]:6.0f} us spin kernel: "
              f"launch -> done {m[2]:7.1f} us -> execution alone {m[2] - m[3]:6.1f} us, start delay of the idle launch {m[1] - (m[2] - m[3]):6.1f} us")
