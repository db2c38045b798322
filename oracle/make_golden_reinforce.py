#!/usr/bin/env python3
"""Generate `tests/golden/reinforce_*.npz` from the REAL reference and pin `oracle/reinforce_oracle.py` against it.

TEST INFRASTRUCTURE.  Runs only in the build container (the reference is mounted read-only at /root/reference).

Drives `recnn.nn.Reinforce(...).update / .step` (algo.py:182-233, update/reinforce.py:69-129) of the reference on seeded
inputs with `torch.optim.Adam` injected through `algo.optimizers[...]`, for the three estimators of `ChooseREINFORCE`.
Two things of the reference run are pinned from outside so that the run is a function of its inputs:
  * `Categorical.sample` (models.py:109,133-134) is replaced by a queue of pre-drawn actions (the reference would draw them
    with torch.multinomial; the HIP path has its own sampler), log_prob and everything downstream is the reference's;
  * `data.get_base_batch` is called with its default device "cuda" in reinforce.py:83 -- redirected to the CPU.
The behaviour policy of the correction estimators is a fixed softmax(state @ Wb) (the notebooks train a `Beta` net there;
any callable `beta(state, action=...) -> probabilities` is legal).

Usage:  python oracle/make_golden_reinforce.py      (from the repo root)
"""
import functools
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("RECNN_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(HERE, "_stubs"))
sys.path.append(ROOT)     # after the reference: `import recnn` must find /root/reference, not the repository's `recnn` shim

import numpy as np  # noqa: E402
import torch  # noqa: E402
from torch.distributions import Categorical  # noqa: E402

import recnn as ref  # noqa: E402  (the reference)

assert ref.__file__.startswith(REF), ref.__file__
from oracle import recnn_oracle as O  # noqa: E402
from oracle import reinforce_oracle as R  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
torch.set_num_threads(1)

_cpu_base_batch = functools.partial(ref.data.utils.get_base_batch, device=torch.device("cpu"))
ref.data.get_base_batch = _cpu_base_batch
ref.data.utils.get_base_batch = _cpu_base_batch

ACTION_QUEUE = []


class QueuedCategorical(Categorical):
    def sample(self, sample_shape=torch.Size()):
        return ACTION_QUEUE.pop(0)


ref.nn.models.Categorical = QueuedCategorical


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def run(name, method, pi_source, S, N, H, B, steps, seed, lr_v, lr_p, wd_v, wd_p, K):
    torch.manual_seed(seed)
    value = ref.nn.Critic(S, N, H, 54e-2)
    policy = ref.nn.DiscreteActor(S, N, H)
    algo = ref.nn.Reinforce(policy, value)
    algo.optimizers["value_optimizer"] = torch.optim.Adam(value.parameters(), lr=lr_v, weight_decay=wd_v)
    algo.optimizers["policy_optimizer"] = torch.optim.Adam(policy.parameters(), lr=lr_p, weight_decay=wd_p)
    Wb = torch.randn(S, N) * 0.3

    def beta(state, action=None):
        return torch.softmax(state @ Wb, dim=1)

    choose = ref.nn.ChooseREINFORCE
    if method == "corr":
        policy.select_action = lambda state, action, K, writer, step, **kw: \
            policy._select_action_with_correction(state, beta, action, writer=writer, step=step)
        algo.params["reinforce"] = choose(choose.reinforce_with_correction)
    elif method == "topk":
        policy.select_action = lambda state, action, K, writer, step, **kw: \
            policy._select_action_with_TopK_correction(state, beta, action, K=K, writer=writer, step=step)
        algo.params["reinforce"] = choose(choose.reinforce_with_TopK_correction)
    algo.params["K"] = K
    policy.action_source = {"pi": pi_source, "beta": "beta"}

    batches = []
    for _ in range(2):
        a = torch.randint(0, N, (B,))
        onehot = torch.zeros(B, N)
        onehot.scatter_(1, a.view(-1, 1), 1)
        batches.append({"state": torch.randn(B, S), "action": onehot, "reward": torch.randn(B) * 3.0,
                        "next_state": torch.randn(B, S), "done": (torch.rand(B) < 0.1).float()})
    pi_draws = torch.randint(0, N, (steps, B))
    beta_draws = torch.randint(0, N, (steps, B))

    ost = R.ReinforceState.create(R.policy_params_from_module(policy), O.params_from_module(value),
                                  R.AdamDict(R.POLICY_ORDER, lr=lr_p, weight_decay=wd_p),
                                  R.AdamDict(O.PARAM_ORDER, lr=lr_v, weight_decay=wd_v), method=method, K=K)
    blob = {f"policy.{k}": v.numpy().copy() for k, v in ost.policy.items()}
    blob.update({f"value.{k}": v.numpy().copy() for k, v in ost.value.items()})
    losses, olosses, lps, olps, all_masks = [], [], [], [], []
    for t in range(steps):
        b = batches[t % 2]
        rng = torch.get_rng_state()
        masks = O.draw_dropout_masks(4, B, H)
        torch.set_rng_state(rng)
        ACTION_QUEUE[:] = [pi_draws[t]] if method == "basic" else [pi_draws[t], beta_draws[t]]
        out = algo.update(b)
        assert not ACTION_QUEUE
        lps.append(policy.saved_log_probs[-1].detach().numpy().copy() if policy.saved_log_probs else None)
        algo.step()
        scored = pi_draws[t] if (method == "basic" or pi_source == "pi") else beta_draws[t]
        oo = R.reinforce_step(ost, {k: v.numpy() for k, v in b.items()}, scored, masks, step=t,
                              beta_probs=None if method == "basic" else beta(b["state"]),
                              beta_action=None if method == "basic" else beta_draws[t])
        olps.append(oo["log_prob"].numpy())
        if out is not None:
            losses.append([t, out["value"], out["policy"]])
            olosses.append([t, oo["value"], oo["policy"]])
        all_masks.append(torch.stack(masks).numpy())
    # log-probs of the steps whose episode was still open when sampled (the update clears the list)
    pairs = [(a, b) for a, b in zip(lps, olps) if a is not None]
    e_lp = max(rel_err(b, a) for a, b in pairs)
    e = rel_err(olosses, losses)
    assert e < 5e-5 and e_lp < 1e-5, (name, "loss", e, e_lp)
    final = {"policy": R.policy_params_from_module(policy), "value": O.params_from_module(value),
             "target_policy": R.policy_params_from_module(algo.nets["target_policy_net"]),
             "target_value": O.params_from_module(algo.nets["target_value_net"])}
    worst = 0.0
    for tag, op in (("policy", ost.policy), ("value", ost.value), ("target_policy", ost.target_policy),
                    ("target_value", ost.target_value)):
        for k, v in final[tag].items():
            ek = rel_err(op[k], v)
            worst = max(worst, ek)
            assert ek < 5e-5, (name, tag, k, ek)
            blob[f"final.{tag}.{k}"] = v.numpy()
    print(f"{name}: {steps} steps, {len(losses)} policy updates; oracle vs reference: loss {e:.2e}, log-prob {e_lp:.2e}, "
          f"params {worst:.2e}")
    for i, b in enumerate(batches):
        blob.update({f"batch{i}.{k}": v.numpy() for k, v in b.items()})
    blob["beta_w"] = Wb.numpy()
    blob["pi_draws"] = pi_draws.numpy()
    blob["beta_draws"] = beta_draws.numpy()
    blob["masks"] = np.stack(all_masks)                    # [steps, 4, B, H] uint8
    blob["losses"] = np.asarray(losses, dtype=np.float64)  # rows: step, value, policy
    blob["hyper"] = np.asarray([lr_v, lr_p, wd_v, wd_p], dtype=np.float64)
    blob["dims"] = np.asarray([S, N, H, B, steps, seed, K])
    blob["method"] = np.asarray(method)
    blob["pi_source"] = np.asarray(pi_source)
    np.savez_compressed(os.path.join(OUT, f"{name}.npz"), **blob)


if __name__ == "__main__":
    run("reinforce_basic", "basic", "pi", S=27, N=40, H=16, B=12, steps=25, seed=11, lr_v=1e-3, lr_p=1e-3, wd_v=1e-2, wd_p=1e-2, K=10)
    run("reinforce_corr", "corr", "pi", S=27, N=40, H=16, B=12, steps=25, seed=12, lr_v=1e-3, lr_p=1e-3, wd_v=1e-2, wd_p=1e-2, K=10)
    run("reinforce_topk", "topk", "beta", S=27, N=44, H=24, B=10, steps=32, seed=13, lr_v=1e-3, lr_p=1e-3, wd_v=0.0, wd_p=1e-2, K=5)
