"""REINFORCE at a 100k-item catalogue: per-step time of recnn_amd.nn.Reinforce.update.
usage: python tools/reinforce_bench.py [--items 100000] [--rows 256] [--hidden 2048] [--steps 31] [--method topk|basic]
Prints one JSON line (median ms of ordinary steps, ms of policy-update steps, it/s over whole policy cycles)."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import recnn_amd  # noqa: E402
from recnn_amd.nn import functional as F_hip  # noqa: E402


def run(items=100000, rows=256, hidden=2048, steps=31, method="topk", optimizer="ranger", dtype="fp32", beta="learned"):
    """One measurement (the body of `main`): returns the record as a dict."""
    a = argparse.Namespace(items=items, rows=rows, hidden=hidden, steps=steps, method=method, optimizer=optimizer, dtype=dtype, beta=beta)
    return _measure(a)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--items", type=int, default=100000)
    ap.add_argument("--rows", type=int, default=256)
    ap.add_argument("--hidden", type=int, default=2048)
    ap.add_argument("--steps", type=int, default=31)
    ap.add_argument("--method", default="topk")
    ap.add_argument("--optimizer", default="ranger")
    ap.add_argument("--dtype", default="fp32", help="compute type of the catalogue GEMMs: fp32 | bf16")
    ap.add_argument("--beta", default="learned", choices=["learned", "frozen"],
                    help="behaviour policy of the Top-K correction: the notebook's Beta net trained inside every step (default) or a frozen projection")
    a = ap.parse_args()
    print(json.dumps(_measure(a)))


def _measure(a):
    recnn_amd.nn.algo.set_default_optimizer(a.optimizer)
    F_hip.set_catalogue_dtype(a.dtype)
    N, S, H, B = a.items, 1290, a.hidden, a.rows
    torch.manual_seed(0)
    value = recnn_amd.nn.Critic(S, N, H, 54e-2).cuda()
    policy = recnn_amd.nn.DiscreteActor(S, N, H).cuda()
    algo = recnn_amd.nn.Reinforce(policy, value).to(torch.device("cuda"))
    beta_ms = []
    if a.method == "topk":
        if a.beta == "learned":
            # the notebook's configuration (3. TopK Reinforce Off Policy Correction.ipynb, cells 3-5): the behaviour policy is a
            # `Beta` net -- Linear(1290, n_items) + softmax -- that takes one optimizer step on its cross entropy inside EVERY call
            beta_net = recnn_amd.nn.Beta(S, N).cuda()

            def beta(state, action=None):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                out = beta_net(state, action)
                e1.record()
                beta_ms.append((e0, e1))
                return out
        else:
            Wb = torch.randn(S, N, device="cuda") * 0.02      # rounds 2-3: a FROZEN random projection (not the notebook's setup)

            def beta(state, action=None):
                return torch.softmax(state @ Wb, dim=1)
        policy.select_action = lambda state, action, K, writer, step, **kw: \
            policy._select_action_with_TopK_correction(state, beta, action, K=K, writer=writer, step=step)
        ch = recnn_amd.nn.ChooseREINFORCE
        algo.params["reinforce"] = ch(ch.reinforce_with_TopK_correction)
        policy.action_source = {"pi": "beta", "beta": "beta"}
    idx = torch.randint(0, N, (B,), device="cuda")
    batch = {"state": torch.randn(B, S, device="cuda"), "action": F_hip.onehot_rows(idx, N), "reward": torch.randn(B, device="cuda"),
             "next_state": torch.randn(B, S, device="cuda"), "done": torch.zeros(B, device="cuda")}
    times, kinds = [], []
    for t in range(a.steps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = algo.update(batch)
        algo.step()
        torch.cuda.synchronize()
        times.append(1e3 * (time.perf_counter() - t0))
        kinds.append(out is not None)
    ordinary = sorted(x for x, k in zip(times[1:], kinds[1:]) if not k)
    pol = [x for x, k in zip(times, kinds) if k]
    cyc = times[11:31] if len(times) >= 31 else times[1:]
    beta_step_ms = sorted(e0.elapsed_time(e1) for e0, e1 in beta_ms[1:])
    return {"n_items": N, "rows": B, "hidden": H, "method": a.method, "optimizer": a.optimizer, "dtype": a.dtype,
            "beta": a.beta if a.method == "topk" else None,
            "beta_train_call_ms": round(beta_step_ms[len(beta_step_ms) // 2], 3) if beta_step_ms else None,
            "ordinary_step_ms": round(ordinary[len(ordinary) // 2], 3), "policy_step_ms": [round(x, 2) for x in pol],
            "it_per_s": round(1e3 * len(cyc) / sum(cyc), 2), "peak_mem_gb": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)}


if __name__ == "__main__":
    main()
