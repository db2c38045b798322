"""CPU tests: the oracle against the committed fixtures generated from the REAL reference
(oracle/make_golden.py), plus the numerical facts the GPU parity method relies on."""
import json
import os

import numpy as np
import torch

from oracle import recnn_oracle as O
from tests.helpers import fro_err, rel_err


def _unpack(g, prefix):
    return {k: torch.from_numpy(g[f"{prefix}.{k}"].copy()) for k in O.PARAM_ORDER}


def test_gather_fixtures_bit_exact(golden_dir):
    for name in ("tiny", "f10e128"):
        g = np.load(os.path.join(golden_dir, f"gather_{name}.npz"))
        lens = g["lengths"]
        offs = np.concatenate([[0], np.cumsum(lens)])
        items = [g["items_flat"][offs[i]:offs[i + 1]] for i in range(len(lens))]
        ratings = [g["ratings_flat"][offs[i]:offs[i + 1]] for i in range(len(lens))]
        out = O.frame_batch(items, ratings, g["table"], int(g["frame"]))
        for k in ("state", "next_state", "action", "reward", "done"):
            assert np.array_equal(out[k], g[k]), (name, k)
        assert np.array_equal(out["sizes"], g["meta_sizes"])
        # structure of the reference's outputs (utils.py:60-71)
        F = int(g["frame"])
        E = g["table"].shape[1]
        assert out["state"].shape[1] == F * E + F
        assert out["done"].sum() == len(lens) and out["done"][np.cumsum(lens - F) - 1].all()


def test_gather_edge_cases():
    tab = np.arange(40, dtype=np.float32).reshape(10, 4)
    items = [np.array([1, 2, 3, 4]), np.array([5, 6, 7, 8, 9])]
    ratings = [np.array([1., 2., 3., 4.]), np.array([-4., -3., 0., 5., 1.])]
    out = O.frame_batch(items, ratings, tab, 3)
    assert out["state"].shape == (3, 15)
    assert np.array_equal(out["state"][0, :12], tab[[1, 2, 3]].reshape(-1))
    assert np.array_equal(out["next_state"][0, :12], tab[[2, 3, 4]].reshape(-1))
    assert np.array_equal(out["action"][1], tab[8]) and out["reward"][2] == 1.0
    assert out["done"].tolist() == [1.0, 0.0, 1.0]
    cut = O.frame_batch(items, ratings, tab, 3, rows=2)
    assert cut["state"].shape[0] == 2 and np.array_equal(cut["state"], out["state"][:2])


def test_ddpg_tiny_fixture(golden_dir):
    g = np.load(os.path.join(golden_dir, "ddpg_tiny.npz"))
    in_dim, act, hid, B, steps, _ = [int(x) for x in g["dims"]]
    lr_v, lr_p, wd_v, wd_p = [float(x) for x in g["hyper"]]
    st = O.DDPGState.create(_unpack(g, "policy"), _unpack(g, "value"), O.AdamState(lr=lr_p, weight_decay=wd_p),
                            O.AdamState(lr=lr_v, weight_decay=wd_v))
    for t in range(steps):
        b = {k: g[f"batch{t % 2}.{k}"] for k in ("state", "action", "reward", "next_state", "done")}
        lo = O.ddpg_step(st, b, [torch.from_numpy(m) for m in g["masks"][t]], step=t)
        assert abs(lo["value"] - g["losses"][t][0]) <= 1e-5 * abs(g["losses"][t][0])
        assert abs(lo["policy"] - g["losses"][t][1]) <= 1e-5 * abs(g["losses"][t][1]) + 1e-7
    for tag, p in (("policy", st.policy), ("value", st.value), ("target_policy", st.target_policy),
                   ("target_value", st.target_value)):
        for k in O.PARAM_ORDER:
            assert rel_err(p[k], g[f"final.{tag}.{k}"]) < 1e-5, (tag, k)


def test_td3_tiny_fixture(golden_dir):
    g = np.load(os.path.join(golden_dir, "td3_tiny.npz"))
    in_dim, act, hid, B, steps, _ = [int(x) for x in g["dims"]]
    lr_v, lr_p, wd_v, wd_p = [float(x) for x in g["hyper"]]
    st = O.TD3State.create(_unpack(g, "policy"), _unpack(g, "value1"), _unpack(g, "value2"),
                           O.AdamState(lr=lr_p, weight_decay=wd_p), O.AdamState(lr=lr_v, weight_decay=wd_v),
                           O.AdamState(lr=lr_v, weight_decay=wd_v))
    for t in range(steps):
        b = {k: g[f"batch{t % 2}.{k}"] for k in ("state", "action", "reward", "next_state", "done")}
        lo = O.td3_step(st, b, torch.from_numpy(g["noise"][t]), [torch.from_numpy(m) for m in g["masks"][t]], step=t)
        for key, r in zip(("value1", "value2", "policy"), g["losses"][t]):
            assert abs(lo[key] - r) <= 1e-5 * abs(r) + 1e-7, (t, key)
    # td3.py:136-141: the target policy is never soft-updated
    for k in O.PARAM_ORDER:
        assert torch.equal(st.target_policy[k], torch.from_numpy(g[f"policy.{k}"]))
        assert rel_err(st.policy[k], g[f"final.policy.{k}"]) < 1e-5
        assert rel_err(st.target_value2[k], g[f"final.target_value2.{k}"]) < 1e-5


def test_ddpg_full_b32_json_recipe(golden_dir):
    js = json.load(open(os.path.join(golden_dir, "ddpg_full_b32.json")))
    S, A, H, B, steps, seed = js["dims"]
    lr_v, lr_p, wd_v, wd_p = js["hyper"]
    torch.manual_seed(seed)

    def mk(inp, out, init_w):  # constructor RNG order of models.py:52-57 / :198-203
        l1, l2, l3 = torch.nn.Linear(inp, H), torch.nn.Linear(H, H), torch.nn.Linear(H, out)
        l3.weight.data.uniform_(-init_w, init_w); l3.bias.data.uniform_(-init_w, init_w)
        return {"w1": l1.weight.data.clone(), "b1": l1.bias.data.clone(), "w2": l2.weight.data.clone(),
                "b2": l2.bias.data.clone(), "w3": l3.weight.data.clone(), "b3": l3.bias.data.clone()}
    val, pol = mk(S + A, 1, 54e-2), mk(S, A, 6e-1)
    batches = [{"state": torch.randn(B, S), "action": torch.randn(B, A), "reward": torch.randn(B) * 3.0,
                "next_state": torch.randn(B, S), "done": (torch.rand(B) < 0.1).float()} for _ in range(2)]
    assert abs(float(batches[0]["state"].double().sum()) - js["input_checksum"][0]) < 1e-6
    st = O.DDPGState.create(pol, val, O.AdamState(lr=lr_p, weight_decay=wd_p), O.AdamState(lr=lr_v, weight_decay=wd_v))
    for t in range(steps):
        masks = O.draw_dropout_masks(6, B, H)
        lo = O.ddpg_step(st, batches[t % 2], masks, step=t)
        assert abs(lo["value"] - js["losses"][t][0]) <= 2e-5 * abs(js["losses"][t][0])
        assert abs(lo["policy"] - js["losses"][t][1]) <= 2e-5 * abs(js["losses"][t][1]) + 1e-7


def test_clip_quirk_matches_torch():
    """clip_grad_norm_(p, -1, 1): L1-normalise and flip the sign (ddpg.py:92)."""
    g = {"a": torch.tensor([0.5, -1.0, 2.0]), "b": torch.tensor([[1.0, -3.0]])}
    ps = [torch.nn.Parameter(torch.zeros_like(v)) for v in g.values()]
    for p, v in zip(ps, g.values()):
        p.grad = v.clone()
    torch.nn.utils.clip_grad_norm_(ps, -1, 1)
    coef = O.clip_grad_quirk_scale(g)
    assert coef < 0
    for p, v in zip(ps, g.values()):
        assert torch.allclose(p.grad, v * coef, rtol=1e-6)


def test_adam_eps_regime_amplifies_roundoff():
    """Why parameters are compared in Frobenius norm (tests/test_gpu_engine.py docstring).

    Two gradient vectors that agree to 1e-6 of max|g| -- the size of fp32 summation-order noise, what a
    thread-count change does to the reference itself -- give Adam(eps=1e-8) updates that differ by a
    sizeable fraction of lr on the elements with |g| ~ eps, while the Frobenius error stays tiny."""
    gen = torch.Generator().manual_seed(0)
    n = 400_000
    g = torch.randn(n, generator=gen) * 1e-3
    g[: n // 100] *= 1e-5                      # a percent of near-dead units, as relu/dropout produce
    noise = torch.randn(n, generator=gen) * 1e-9
    outs = []
    w0 = torch.randn(n, generator=gen) * 0.02      # nn.Linear-sized weights
    for grad in (g, g + noise):
        p = {k: torch.zeros(1) for k in O.PARAM_ORDER}
        p["w1"] = w0.clone()
        grads = {k: torch.zeros(1) for k in O.PARAM_ORDER}
        grads["w1"] = grad
        O.adam_step(p, grads, O.AdamState(lr=1e-3))
        outs.append(p["w1"])
    assert rel_err(g + noise, g) < 2e-6
    worst = float((outs[0] - outs[1]).abs().max())
    assert worst > 0.02 * 1e-3                 # > 2% of lr on some element
    assert fro_err(outs[0], outs[1]) < 1e-3     # while the Frobenius error of the parameters stays small


def test_reinforce_oracle_against_reference_fixtures(golden_dir):
    """oracle/reinforce_oracle.py replays the three estimator runs of the real reference (basic, off-policy correction,
    top-K correction with the behaviour policy's action) from their inputs: losses and all four networks agree."""
    from tests import reinforce_replay as RR
    for name in ("reinforce_basic", "reinforce_corr", "reinforce_topk"):
        fx = RR.load(os.path.join(golden_dir, name + ".npz"))
        losses, final = RR.replay_oracle(fx)
        ref = fx["g"]["losses"]
        assert losses.shape == ref.shape and np.array_equal(losses[:, 0], ref[:, 0])
        assert rel_err(losses[:, 1:], ref[:, 1:]) < 5e-5, name
        for tag, p in final.items():
            for k, v in p.items():
                assert rel_err(v, fx["g"][f"final.{tag}.{k}"]) < 5e-5, (name, tag, k)


def test_reinforce_returns_and_estimator_gradients():
    """The hand-written d loss / d log_prob of the three estimators equals autograd's (the correction weight is not
    detached in the reference), and the return normalisation follows reinforce.py:45-53."""
    from oracle import reinforce_oracle as R
    torch.manual_seed(3)
    T, B, K = 5, 7, 4
    rewards = [torch.randn(()) for _ in range(T)]
    ret = R.discounted_returns(rewards)
    run, raw = 0.0, []
    for r in reversed(rewards):
        run = float(r) + 0.99 * run
        raw.insert(0, run)
    raw = np.asarray(raw)
    assert np.allclose(ret.numpy(), (raw - raw.mean()) / (raw.std(ddof=1) + 1e-4), rtol=1e-5, atol=1e-6)
    for method in ("basic", "corr", "topk"):
        lps = [(-torch.rand(B) * 3).requires_grad_() for _ in range(T)]
        blps = [-torch.rand(B) * 3 for _ in range(T)]
        loss, glps = R.reinforce_loss(method, lps, blps, ret, K)
        auto = torch.autograd.grad(loss, lps)
        for a, g in zip(auto, glps):
            assert torch.allclose(a, g.detach(), rtol=1e-5, atol=1e-6), method


def test_bcq_oracle_reproduces_the_reference_run(golden_dir):
    """tests/golden/bcq_small.npz: 9 bcq_update steps of the real reference (F re-pointed at torch.nn.functional) -- the
    oracle restatement reproduces losses to 5e-5 and every stored final-parameter sample to 1e-4; the modules of
    recnn_amd built under the fixture's seed start from the reference's initial weights (checksums)."""
    import os
    from tests import bcq_replay as BR
    fx = BR.load(os.path.join(golden_dir, "bcq_small.npz"))
    losses, final = BR.replay_oracle(fx)
    ref = fx["g"]["losses"]
    for j in range(3):
        assert np.abs(losses[:, j] - ref[:, j]).max() <= 5e-5 * np.abs(ref[:, j]).max(), j
    worst = BR.compare_final(fx, final, rtol=1e-4)
    assert worst < 1e-4


def test_beta_oracle_matches_the_notebooks_class(golden_dir):
    """oracle.reinforce_oracle.beta_step against the fixture made by exec()ing the reference notebook's own `Beta` class
    (oracle/make_golden_beta.py): the probabilities each call returns and the parameters after 8 training calls."""
    import os
    import numpy as np
    import torch
    from oracle import reinforce_oracle as R
    g = np.load(os.path.join(golden_dir, "beta_net.npz"))
    S, N, B, steps, _ = (int(x) for x in g["dims"])
    lr, wd = (float(x) for x in g["hyper"])
    p = {"w": torch.from_numpy(g["w0"].copy()), "b": torch.from_numpy(g["b0"].copy())}
    opt = R.AdamDict(("w", "b"), lr=lr, weight_decay=wd)
    for t in range(steps):
        probs, loss = R.beta_step(p, opt, torch.from_numpy(g["states"][t]), torch.from_numpy(g["targets"][t]))
        assert float((probs - torch.from_numpy(g["probs"][t])).abs().max() / g["probs"][t].max()) < 2e-6, t
        assert abs(loss - float(g["losses"][t])) < 1e-6
    for k in ("w", "b"):
        ref = torch.from_numpy(g["final_" + k])
        assert float((p[k] - ref).abs().max() / ref.abs().max()) < 5e-5, k
