"""Golden runs of the REAL reference SAC step for tests/golden/sac_*.npz.  Runs only in the build container: it reads
`/root/reference/examples/1. Vanilla RL/4. SAC.ipynb` and exec()s the notebook's own code cells (5: StateCritic, 6: SoftQ,
7: StochasticActor, 8: soft_q_update) -- nothing of them is copied into this repository -- against stand-ins for the
notebook's globals (CPU device, a no-op debugger / writer, torch.optim.Adam where the notebook takes torch_optimizer.RAdam,
which is not installed).  The policy's dropout masks are logged by drawing them with the RNG state the notebook's nn.Dropout
is about to consume, its scalar z draws by wrapping `normal_dist.sample`.  While generating, the run is replayed through
oracle/sac_oracle.py and must agree at 5e-5.

usage: python oracle/make_golden_sac.py
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
sys.path.insert(0, os.path.join(ROOT, "oracle", "_stubs"))     # `import torch_optimizer` of recnn/nn/algo.py:6 (package absent)
from oracle import recnn_oracle as O       # noqa: E402
from oracle import sac_oracle as SO        # noqa: E402

NB = "/root/reference/examples/1. Vanilla RL/4. SAC.ipynb"
OUT = os.path.join(ROOT, "tests", "golden")


def notebook_namespace():
    import recnn                      # the reference package (get_base_batch, plot helpers)
    import torch.nn as nn
    import torch.nn.functional as F
    cells = [c for c in json.load(open(NB))["cells"]]

    class _Sink:
        def __getattr__(self, name):
            return lambda *a, **k: None
    ns = {"torch": torch, "nn": nn, "F": F, "np": np, "recnn": recnn, "cuda": torch.device("cpu"), "debugger": _Sink(),
          "writer": _Sink()}
    for idx in (5, 6, 7, 8):          # class StateCritic / SoftQ / StochasticActor, def soft_q_update
        exec("".join(cells[idx]["source"]), ns)
    # recnn.data.get_base_batch defaults to device cuda: give the notebook function CPU batches through a wrapper
    base = recnn.data.get_base_batch
    ns["recnn"] = type("R", (), {"data": type("D", (), {"get_base_batch": staticmethod(lambda b, **k: base(b, device=torch.device("cpu")))}),
                                 "plot": recnn.plot if hasattr(recnn, "plot") else None})
    return ns


def rel_err(a, b):
    a, b = torch.as_tensor(np.asarray(a), dtype=torch.float64), torch.as_tensor(np.asarray(b), dtype=torch.float64)
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def run(name, S, A, H, B, steps, seed, lr, wd):
    ns = notebook_namespace()
    torch.manual_seed(seed)
    params = {"gamma": 0.99, "soft_tau": 0.001, "mean_lambda": 1e-3, "std_lambda": 1e-3, "z_lambda": 1e-10}
    ap = {"mean_initw": 1e-1, "std_initw": 6e-1, "log_std_min": -2, "log_std_max": 2}
    value_net, target_value_net = ns["StateCritic"](S, H, 1e-1), ns["StateCritic"](S, H)
    soft_q_net = ns["SoftQ"](S, A, H, 2e-1)
    policy_net = ns["StochasticActor"](S, A, H, ap)
    for tp, p in zip(target_value_net.parameters(), value_net.parameters()):
        tp.data.copy_(p.data)
    ns.update(value_net=value_net, target_value_net=target_value_net, soft_q_net=soft_q_net, policy_net=policy_net,
              value_criterion=torch.nn.MSELoss(), soft_q_criterion=torch.nn.MSELoss(),
              value_optimizer=torch.optim.Adam(value_net.parameters(), lr=lr, weight_decay=wd),
              soft_q_optimizer=torch.optim.Adam(soft_q_net.parameters(), lr=lr, weight_decay=wd),
              policy_optimizer=torch.optim.Adam(policy_net.parameters(), lr=lr, weight_decay=wd),
              soft_update=lambda net, tgt, soft_tau=1e-2: [tp.data.copy_(tp.data * (1.0 - soft_tau) + p.data * soft_tau)
                                                           for tp, p in zip(tgt.parameters(), net.parameters())])
    zs = []
    real_sample = policy_net.normal_dist.sample

    def sample():
        z = real_sample()
        zs.append(float(z))
        return z
    policy_net.normal_dist.sample = sample

    batches = []
    for _ in range(2):
        batches.append({"state": torch.randn(B, S), "action": torch.randn(B, A) * 0.5, "reward": torch.randn(B) * 3.0,
                        "next_state": torch.randn(B, S), "done": (torch.rand(B) < 0.1).float()})
    ost = SO.SACState(value=SO.snapshot(SO.critic_params_from_module(value_net)),
                      target_value=SO.snapshot(SO.critic_params_from_module(target_value_net)),
                      soft_q=SO.snapshot(SO.critic_params_from_module(soft_q_net)),
                      policy=SO.snapshot(SO.policy_params_from_module(policy_net)),
                      value_opt=SO.Adam(lr=lr, weight_decay=wd), soft_q_opt=SO.Adam(lr=lr, weight_decay=wd),
                      policy_opt=SO.Adam(lr=lr, weight_decay=wd), params=dict(params, **ap))
    blob = {}
    for tag, p in (("value", ost.value), ("soft_q", ost.soft_q), ("policy", ost.policy)):
        blob.update({f"{tag}.{k}": v.numpy().copy() for k, v in p.items()})
    losses, olosses, all_masks, worst_lp = [], [], [], 0.0
    import warnings
    for t in range(steps):
        b = batches[t % 2]
        rng = torch.get_rng_state()
        masks = O.draw_dropout_masks(2, B, H)            # what the policy's two nn.Dropout calls are about to draw
        torch.set_rng_state(rng)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")              # MSELoss broadcast warning ([B,1] vs [B,A]) is the notebook's own
            out = ns["soft_q_update"](t, b, params, learn=True)
        oo = SO.sac_step(ost, b, zs[-1], masks, step=t)
        losses.append([t, out["value"], out["softq"], out["policy"]])
        olosses.append([t, oo["value"], oo["softq"], oo["policy"]])
        all_masks.append(torch.stack(masks).numpy())
    e = rel_err(olosses, losses)
    assert e < 5e-5, (name, "losses", e)
    worst = 0.0
    for tag, net, snap, op in (("value", value_net, SO.critic_params_from_module, ost.value),
                               ("target_value", target_value_net, SO.critic_params_from_module, ost.target_value),
                               ("soft_q", soft_q_net, SO.critic_params_from_module, ost.soft_q),
                               ("policy", policy_net, SO.policy_params_from_module, ost.policy)):
        for k, v in SO.snapshot(snap(net)).items():
            ek = rel_err(op[k], v)
            worst = max(worst, ek)
            assert ek < 5e-5, (name, tag, k, ek)
            blob[f"final.{tag}.{k}"] = v.numpy()
    print(f"{name}: {steps} steps; oracle vs the notebook's cells: losses {e:.2e}, params {worst:.2e}")
    for i, b in enumerate(batches):
        blob.update({f"batch{i}.{k}": v.numpy() for k, v in b.items()})
    blob["z"] = np.asarray(zs, dtype=np.float64)
    blob["masks"] = np.stack(all_masks)                   # [steps, 2, B, H] uint8
    blob["losses"] = np.asarray(losses, dtype=np.float64)  # rows: step, value, softq, policy
    blob["dims"] = np.asarray([S, A, H, B, steps, seed])
    blob["hyper"] = np.asarray([lr, wd], dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, f"{name}.npz"), **blob)


if __name__ == "__main__":
    run("sac_small", S=27, A=8, H=16, B=12, steps=20, seed=21, lr=1e-3, wd=0.0)
    run("sac_wd", S=33, A=6, H=24, B=10, steps=16, seed=22, lr=3e-4, wd=1e-2)
