#!/usr/bin/env python3
"""Golden runs of the reference's learned behaviour policy `Beta` for tests/golden/beta_net.npz.

TEST INFRASTRUCTURE.  Runs only in the build container: it reads
`/root/reference/examples/2. REINFORCE TopK Off Policy Correction/3. TopK Reinforce Off Policy Correction.ipynb` and exec()s the
notebook's own cell 3 (`class Beta`) -- nothing of it is copied into this repository -- against stand-ins for the notebook's globals
(`num_items`; `optim.RAdam` -> torch.optim.Adam with the same lr / weight_decay: torch_optimizer is not installed and un-pinned,
as for the SAC fixture).  The notebook's `nn.Softmax()` has no dim (dim 1 for 2-D input, with a deprecation warning).
While generating, every call is replayed through oracle/reinforce_oracle.py::beta_step and must agree (probabilities 2e-6, final parameters 5e-5).

usage: python oracle/make_golden_beta.py
"""
import json
import os
import sys
import types
import warnings

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import reinforce_oracle as R       # noqa: E402

NB = "/root/reference/examples/2. REINFORCE TopK Off Policy Correction/3. TopK Reinforce Off Policy Correction.ipynb"
OUT = os.path.join(ROOT, "tests", "golden")
torch.set_num_threads(1)


def notebook_beta(num_items, lr_seen):
    import torch.nn as nn
    cells = json.load(open(NB))["cells"]
    src = "".join(cells[3]["source"])
    assert src.lstrip().startswith("class Beta(nn.Module):"), src[:80]

    def radam(params, lr, weight_decay):
        lr_seen.append((lr, weight_decay))
        return torch.optim.Adam(params, lr=lr, weight_decay=weight_decay)
    ns = {"torch": torch, "nn": nn, "num_items": num_items, "optim": types.SimpleNamespace(RAdam=radam)}
    exec(src, ns)
    return ns["Beta"]


def rel_err(a, b):
    a, b = torch.as_tensor(np.asarray(a), dtype=torch.float64), torch.as_tensor(np.asarray(b), dtype=torch.float64)
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def run(name, N, B, steps, seed, lr=None):
    seen = []
    Beta = notebook_beta(N, seen)
    torch.manual_seed(seed)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        net = Beta()
        assert seen == [(1e-5, 1e-5)], seen                      # the notebook's own hyper-parameters
        if lr is not None:                                        # a learning rate at which 8 steps are visible in fp32
            for g in net.optim.param_groups:
                g["lr"] = lr
        S = net.net[0].in_features
        assert S == 1290
        p = {"w": net.net[0].weight.detach().clone(), "b": net.net[0].bias.detach().clone()}
        blob = {"w0": p["w"].numpy().copy(), "b0": p["b"].numpy().copy()}
        opt = R.AdamDict(("w", "b"), lr=net.optim.param_groups[0]["lr"], weight_decay=1e-5)
        worst = worst_p = 0.0
        states, targets, probs_all, losses = [], [], [], []
        for t in range(steps):
            state = torch.randn(B, S)
            tgt = torch.randint(0, N, (B,))
            action = torch.zeros(B, N).scatter_(1, tgt.view(-1, 1), 1.0)
            got = net(state, action)                              # trains, returns the probabilities before the step
            want, loss = R.beta_step(p, opt, state, tgt)
            worst = max(worst, rel_err(want, got))
            states.append(state.numpy()); targets.append(tgt.numpy()); probs_all.append(got.numpy().copy()); losses.append(loss)
        for k, v in (("w", net.net[0].weight), ("b", net.net[0].bias)):
            e = rel_err(p[k], v.detach())
            worst_p = max(worst_p, e)
            blob["final_" + k] = v.detach().numpy().copy()
    # probabilities: fp32 round-off; parameters after 8 Adam steps at lr = 1e-3: the first steps move every element by ~lr sign(g),
    # elements whose gradient is at round-off level can land 2 lr apart (as in the other fixtures: 5e-5 of the tensor's scale)
    assert worst < 2e-6 and worst_p < 5e-5, (worst, worst_p)
    moved = float((torch.from_numpy(blob["final_w"]) - torch.from_numpy(blob["w0"])).abs().max())
    print(f"{name}: {steps} calls of the notebook's Beta (N={N}, B={B}); oracle vs notebook: probabilities {worst:.2e}, final parameters {worst_p:.2e}; weights moved by {moved:.2e}")
    blob.update(states=np.stack(states), targets=np.stack(targets), probs=np.stack(probs_all), losses=np.asarray(losses),
                dims=np.asarray([1290, N, B, steps, seed]), hyper=np.asarray([opt.lr, 1e-5]))
    np.savez_compressed(os.path.join(OUT, f"{name}.npz"), **blob)


if __name__ == "__main__":
    run("beta_net", N=48, B=12, steps=8, seed=21, lr=1e-4)
