#!/usr/bin/env python3
"""Generate `tests/golden/*` from the REAL reference and pin the oracle against it.

TEST INFRASTRUCTURE.  Runs only in the build container, where the upstream reference is
mounted read-only at /root/reference (it does not exist on the GPU box).  It

  1. imports the reference package `recnn` from /root/reference behind the
     `oracle/_stubs/torch_optimizer` import stub (reference `recnn/nn/algo.py:6`),
  2. drives the reference's own functions on seeded inputs:
       - `recnn.data.utils.prepare_batch_static_size`  (utils.py:161-187, :51-81)
       - `recnn.nn.DDPG(...).update / .step`            (algo.py:43-62, ddpg.py:8-104)
       - `recnn.nn.TD3(...).update / .step`             (td3.py:8-150)
     with `torch.optim.Adam` injected through the public `algo.optimizers[...]` dict,
  3. checks `oracle/recnn_oracle.py` against those outputs (asserts), and
  4. writes the fixtures the CPU and GPU test-suites load.

Usage:  python oracle/make_golden.py            (from the repo root)
"""
import json
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

import recnn as ref  # noqa: E402  (the reference)

assert ref.__file__.startswith(REF), ref.__file__
from oracle import recnn_oracle as O  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
torch.set_num_threads(1)  # fixtures are single-thread CPU results


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


# ------------------------------------------------------------------ gather fixtures
def make_users(rng, lengths, n_items):
    items = [rng.integers(0, n_items, size=L).astype(np.int64) for L in lengths]
    ratings = [(2.0 * (rng.integers(1, 11, size=L) * 0.5 - 2.5)).astype(np.float64) for L in lengths]  # 2*(r-2.5)
    return items, ratings


def gather_case(name, lengths, n_items, emb_dim, frame, seed):
    rng = np.random.default_rng(seed)
    items, ratings = make_users(rng, lengths, n_items)
    table = torch.from_numpy(rng.standard_normal((n_items, emb_dim)).astype(np.float32))
    users = [int(100 + 7 * i) for i in range(len(lengths))]
    batch_list = [{"items": items[i], "rates": ratings[i], "sizes": len(items[i]), "users": users[i]}
                  for i in range(len(lengths))]
    out = ref.data.utils.prepare_batch_static_size(batch_list, table, frame_size=frame)
    mine = O.frame_batch(items, ratings, table.numpy(), frame)
    for k in ("state", "action", "reward", "next_state", "done"):
        assert np.array_equal(out[k].numpy(), mine[k]), (name, k)
    assert np.array_equal(out["meta"]["sizes"].numpy(), mine["sizes"])
    np.savez_compressed(
        os.path.join(OUT, f"gather_{name}.npz"),
        lengths=np.asarray(lengths), frame=frame, users=np.asarray(users),
        items_flat=np.concatenate(items), ratings_flat=np.concatenate(ratings), table=table.numpy(),
        state=out["state"].numpy(), action=out["action"].numpy(), reward=out["reward"].numpy(),
        next_state=out["next_state"].numpy(), done=out["done"].numpy(),
        meta_sizes=out["meta"]["sizes"].numpy(), meta_users=out["meta"]["users"].numpy())
    print(f"gather_{name}: B={out['state'].shape[0]} bit-exact vs oracle")


# ------------------------------------------------------------------ update fixtures
def with_masks(n_masks, B, H, noise_shape=None, noise_std=None):
    """Pre-draw exactly what the reference's next step will draw, then rewind the RNG."""
    st = torch.get_rng_state()
    noise = O.draw_td3_noise(noise_shape[0], noise_shape[1], noise_std) if noise_shape else None
    masks = O.draw_dropout_masks(n_masks, B, H)
    torch.set_rng_state(st)
    return noise, masks


def snap(mod):
    return O.params_from_module(mod)


def pack(prefix, p):
    return {f"{prefix}.{k}": v.numpy().copy() for k, v in p.items()}   # copy: the oracle updates in place


def check_params(tag, oracle_p, module, tol):
    rp = snap(module)
    for k in O.PARAM_ORDER:
        e = rel_err(oracle_p[k], rp[k])
        assert e < tol, (tag, k, e)


def run_ddpg(name, in_dim, act_dim, hid, B, steps, seed, lr_v, lr_p, wd_v, wd_p, store_full):
    torch.manual_seed(seed)
    value = ref.nn.Critic(in_dim, act_dim, hid, 54e-2)
    policy = ref.nn.Actor(in_dim, act_dim, hid, 6e-1)
    algo = ref.nn.DDPG(policy, value)
    algo.optimizers["value_optimizer"] = torch.optim.Adam(value.parameters(), lr=lr_v, weight_decay=wd_v)
    algo.optimizers["policy_optimizer"] = torch.optim.Adam(policy.parameters(), lr=lr_p, weight_decay=wd_p)
    batches = []
    for _ in range(2):
        batches.append({
            "state": torch.randn(B, in_dim), "action": torch.randn(B, act_dim),
            "reward": torch.randn(B) * 3.0, "next_state": torch.randn(B, in_dim),
            "done": (torch.rand(B) < 0.1).float()})
    ost = O.DDPGState.create(snap(policy), snap(value),
                             O.AdamState(lr=lr_p, weight_decay=wd_p), O.AdamState(lr=lr_v, weight_decay=wd_v))
    init = {**pack("policy", ost.policy), **pack("value", ost.value)}
    losses, olosses, all_masks = [], [], []
    for t in range(steps):
        b = batches[t % 2]
        _, masks = with_masks(6, B, hid)
        lr_ = algo.update(b, learn=True)
        algo.step()
        lo = O.ddpg_step(ost, {k: v.numpy() for k, v in b.items()}, masks, step=t, learn=True)
        losses.append([lr_["value"], lr_["policy"]])
        olosses.append([lo["value"], lo["policy"]])
        all_masks.append(torch.stack(masks).numpy())
    e = rel_err(olosses, losses)
    assert e < 2e-5, (name, "loss", e)
    for tag, op, mod in (("policy", ost.policy, policy), ("value", ost.value, value),
                         ("target_policy", ost.target_policy, algo.nets["target_policy_net"]),
                         ("target_value", ost.target_value, algo.nets["target_value_net"])):
        check_params(name + ":" + tag, op, mod, 2e-5)
    print(f"{name}: {steps} steps, oracle vs reference loss rel-err {e:.2e}")
    final = {}
    for tag, mod in (("policy", policy), ("value", value), ("target_policy", algo.nets["target_policy_net"]),
                     ("target_value", algo.nets["target_value_net"])):
        final[tag] = snap(mod)
    if store_full:
        blob = dict(init)
        for tag, p in final.items():
            blob.update(pack("final." + tag, p))
        for i, b in enumerate(batches):
            blob.update({f"batch{i}.{k}": v.numpy() for k, v in b.items()})
        blob["masks"] = np.stack(all_masks)          # [steps, 6, B, H] uint8
        blob["losses"] = np.asarray(losses, dtype=np.float64)
        blob["hyper"] = np.asarray([lr_v, lr_p, wd_v, wd_p], dtype=np.float64)
        blob["dims"] = np.asarray([in_dim, act_dim, hid, B, steps, seed])
        np.savez_compressed(os.path.join(OUT, f"{name}.npz"), **blob)
    else:
        js = {"dims": [in_dim, act_dim, hid, B, steps, seed], "hyper": [lr_v, lr_p, wd_v, wd_p],
              "losses": losses,
              "input_checksum": [float(batches[0]["state"].double().sum()), float(batches[1]["next_state"].double().sum())],
              "init_checksum": {k: float(np.abs(v).astype(np.float64).sum()) for k, v in init.items()},
              "final_abs_sum": {tag: {k: float(v.double().abs().sum()) for k, v in p.items()} for tag, p in final.items()},
              "final_sample": {tag: {k: [float(x) for x in v.flatten()[:: max(1, v.numel() // 7)][:8]] for k, v in p.items()}
                               for tag, p in final.items()},
              "recipe": "torch.manual_seed(seed); Critic(in,act,hid,54e-2); Actor(in,act,hid,6e-1); "
                        "2 batches of randn/randn/randn*3/randn/(rand<0.1); per step: 6 masks "
                        "torch.empty(B,hid).bernoulli_(0.5) drawn from the global generator; "
                        "batch t%2; torch.set_num_threads(1)"}
        with open(os.path.join(OUT, f"{name}.json"), "w") as f:
            json.dump(js, f, indent=1)


def run_td3(name, in_dim, act_dim, hid, B, steps, seed, lr_v, lr_p, wd_v, wd_p):
    torch.manual_seed(seed)
    value1 = ref.nn.Critic(in_dim, act_dim, hid, 54e-2)
    value2 = ref.nn.Critic(in_dim, act_dim, hid, 54e-2)
    policy = ref.nn.Actor(in_dim, act_dim, hid, 6e-1)
    algo = ref.nn.TD3(policy, value1, value2)
    algo.optimizers["value_optimizer1"] = torch.optim.Adam(value1.parameters(), lr=lr_v, weight_decay=wd_v)
    algo.optimizers["value_optimizer2"] = torch.optim.Adam(value2.parameters(), lr=lr_v, weight_decay=wd_v)
    algo.optimizers["policy_optimizer"] = torch.optim.Adam(policy.parameters(), lr=lr_p, weight_decay=wd_p)
    batches = []
    for _ in range(2):
        batches.append({
            "state": torch.randn(B, in_dim), "action": torch.randn(B, act_dim),
            "reward": torch.randn(B) * 3.0, "next_state": torch.randn(B, in_dim),
            "done": (torch.rand(B) < 0.1).float()})
    ost = O.TD3State.create(snap(policy), snap(value1), snap(value2),
                            O.AdamState(lr=lr_p, weight_decay=wd_p), O.AdamState(lr=lr_v, weight_decay=wd_v),
                            O.AdamState(lr=lr_v, weight_decay=wd_v))
    init = {**pack("policy", ost.policy), **pack("value1", ost.value1), **pack("value2", ost.value2)}
    losses, olosses, all_masks, all_noise = [], [], [], []
    std = algo.params["noise_std"]
    for t in range(steps):
        b = batches[t % 2]
        noise, masks = with_masks(8, B, hid, (B, act_dim), std)
        lr_ = algo.update(b, learn=True)
        algo.step()
        lo = O.td3_step(ost, {k: v.numpy() for k, v in b.items()}, noise, masks, step=t, learn=True)
        losses.append([lr_["value1"], lr_["value2"], lr_["policy"]])
        olosses.append([lo["value1"], lo["value2"], lo["policy"]])
        all_masks.append(torch.stack(masks).numpy())
        all_noise.append(noise.numpy())
    e = rel_err(olosses, losses)
    assert e < 2e-5, (name, "loss", e)
    nets = {"policy": policy, "value1": value1, "value2": value2,
            "target_policy": algo.nets["target_policy_net"], "target_value1": algo.nets["target_value_net1"],
            "target_value2": algo.nets["target_value_net2"]}
    for tag, op in (("policy", ost.policy), ("value1", ost.value1), ("value2", ost.value2),
                    ("target_policy", ost.target_policy), ("target_value1", ost.target_value1),
                    ("target_value2", ost.target_value2)):
        check_params(name + ":" + tag, op, nets[tag], 2e-5)
    print(f"{name}: {steps} steps, oracle vs reference loss rel-err {e:.2e}")
    blob = dict(init)
    for tag, mod in nets.items():
        blob.update(pack("final." + tag, snap(mod)))
    for i, b in enumerate(batches):
        blob.update({f"batch{i}.{k}": v.numpy() for k, v in b.items()})
    blob["masks"] = np.stack(all_masks)
    blob["noise"] = np.stack(all_noise)
    blob["losses"] = np.asarray(losses, dtype=np.float64)
    blob["hyper"] = np.asarray([lr_v, lr_p, wd_v, wd_p], dtype=np.float64)
    blob["dims"] = np.asarray([in_dim, act_dim, hid, B, steps, seed])
    np.savez_compressed(os.path.join(OUT, f"{name}.npz"), **blob)


def check_rng_recipes():
    """The two RNG identities the parity tests rely on."""
    torch.manual_seed(3)
    x = torch.randn(5, 16)
    st = torch.get_rng_state()
    y = torch.nn.Dropout(0.5)(x)
    torch.set_rng_state(st)
    m = torch.empty(5, 16).bernoulli_(0.5)
    assert torch.equal(y, x * (m / 0.5)) and torch.equal(y, x * (m * 2.0))
    torch.manual_seed(4)
    a = torch.normal(torch.zeros(7, 9), 0.5)
    torch.manual_seed(4)
    b = torch.randn(7, 9) * 0.5
    assert torch.equal(a, b)
    # clip_grad_norm_(.., -1, 1) quirk (ddpg.py:92)
    p = torch.nn.Parameter(torch.zeros(3))
    p.grad = torch.tensor([0.5, -1.0, 2.0])
    torch.nn.utils.clip_grad_norm_([p], -1, 1)
    coef = O.clip_grad_quirk_scale({"g": torch.tensor([0.5, -1.0, 2.0])})
    assert torch.allclose(p.grad, torch.tensor([0.5, -1.0, 2.0]) * coef, rtol=1e-6), (p.grad, coef)
    print("rng / clip recipes verified:", p.grad.tolist())


if __name__ == "__main__":
    check_rng_recipes()
    gather_case("tiny", [4, 5, 9, 4, 13], n_items=20, emb_dim=8, frame=3, seed=11)
    gather_case("f10e128", [11, 12, 20], n_items=64, emb_dim=128, frame=10, seed=12)
    run_ddpg("ddpg_tiny", in_dim=27, act_dim=8, hid=16, B=10, steps=12, seed=5,
             lr_v=1e-3, lr_p=3e-3, wd_v=0.0, wd_p=1e-2, store_full=True)
    run_td3("td3_tiny", in_dim=27, act_dim=8, hid=16, B=10, steps=12, seed=6,
            lr_v=1e-3, lr_p=3e-3, wd_v=1e-2, wd_p=0.0)
    run_ddpg("ddpg_full_b32", in_dim=1290, act_dim=128, hid=256, B=32, steps=12, seed=0,
             lr_v=1e-3, lr_p=1e-3, wd_v=0.0, wd_p=0.0, store_full=False)
    print("fixtures written to", OUT)
