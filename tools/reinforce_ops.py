"""Which torch ops put ATen kernels into the 100k-item REINFORCE step: torch.profiler over 11 steps (one policy
update), device time grouped by op + input shapes.  usage: python tools/reinforce_ops.py [--dtype bf16]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import recnn_amd  # noqa: E402
from recnn_amd.nn import functional as F_hip  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--stack", action="store_true")
    a = ap.parse_args()
    recnn_amd.nn.algo.set_default_optimizer("ranger")
    F_hip.set_catalogue_dtype(a.dtype)
    N, S, H, B = 100000, 1290, 2048, 256
    torch.manual_seed(0)
    value = recnn_amd.nn.Critic(S, N, H, 54e-2).cuda()
    policy = recnn_amd.nn.DiscreteActor(S, N, H).cuda()
    algo = recnn_amd.nn.Reinforce(policy, value).to(torch.device("cuda"))
    beta_net = recnn_amd.nn.Beta(S, N).cuda()
    policy.select_action = lambda state, action, K, writer, step, **kw: \
        policy._select_action_with_TopK_correction(state, beta_net, action, K=K, writer=writer, step=step)
    ch = recnn_amd.nn.ChooseREINFORCE
    algo.params["reinforce"] = ch(ch.reinforce_with_TopK_correction)
    policy.action_source = {"pi": "beta", "beta": "beta"}
    idx = torch.randint(0, N, (B,), device="cuda")
    batch = {"state": torch.randn(B, S, device="cuda"), "action": F_hip.onehot_rows(idx, N), "reward": torch.randn(B, device="cuda"),
             "next_state": torch.randn(B, S, device="cuda"), "done": torch.zeros(B, device="cuda")}
    for t in range(11):                       # warm: steps 0..10 (policy update at 10)
        algo.update(batch); algo.step()
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=a.stack) as prof:
        for t in range(10):                   # steps 11..20 (policy update at 20)
            algo.update(batch); algo.step()
        torch.cuda.synchronize()
    ka = prof.key_averages(group_by_input_shape=True)
    rows = sorted(ka, key=lambda e: -getattr(e, "self_device_time_total", getattr(e, "self_cuda_time_total", 0)))
    tot = sum(getattr(e, "self_device_time_total", getattr(e, "self_cuda_time_total", 0)) for e in ka)
    print("total device time %.1f ms over 10 steps" % (tot / 1e3))
    for e in rows[:40]:
        t = getattr(e, "self_device_time_total", getattr(e, "self_cuda_time_total", 0))
        if t <= 0:
            continue
        print("%-46s n=%4d %8.1f us %5.1f%%  %s" % (e.key[:46], e.count, t, 100 * t / tot, str(e.input_shapes)[:110]))
    if a.stack:
        print("---- by python stack (aten ops only)")
        ks = prof.key_averages(group_by_stack_n=6)
        rows = sorted((e for e in ks if e.key.startswith("aten::")), key=lambda e: -getattr(e, "self_device_time_total", 0))
        for e in rows[:24]:
            t = getattr(e, "self_device_time_total", 0)
            if t <= 0:
                continue
            frames = [f for f in e.stack if "recnn_amd" in f or "tools/" in f][:3]
            print("%-18s n=%4d %8.1f us | %s" % (e.key[:18], e.count, t, " <- ".join(x.strip()[-70:] for x in frames)))


if __name__ == "__main__":
    main()
