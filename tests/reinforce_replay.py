"""Replays a `tests/golden/reinforce_*.npz` fixture (a run of the REAL reference, oracle/make_golden_reinforce.py) through
either the CPU oracle or the HIP product path; shared by the CPU and GPU tests."""
import numpy as np
import torch


def load(path):
    g = np.load(path)
    S, N, H, B, steps, seed, K = (int(x) for x in g["dims"])
    lr_v, lr_p, wd_v, wd_p = (float(x) for x in g["hyper"])
    return dict(g=g, S=S, N=N, H=H, B=B, steps=steps, K=K, lr_v=lr_v, lr_p=lr_p, wd_v=wd_v, wd_p=wd_p,
                method=str(g["method"]), pi_source=str(g["pi_source"]))


def batches(fx, device=None):
    out = []
    for i in range(2):
        b = {k: torch.from_numpy(fx["g"][f"batch{i}.{k}"].copy()) for k in ("state", "action", "reward", "next_state", "done")}
        out.append({k: v.to(device) for k, v in b.items()} if device is not None else b)
    return out


def beta_fn(fx, device=None):
    Wb = torch.from_numpy(fx["g"]["beta_w"].copy())
    Wb = Wb.to(device) if device is not None else Wb

    def beta(state, action=None):
        return torch.softmax(state @ Wb, dim=1)
    return beta


def replay_oracle(fx, learned_beta=False):
    """learned_beta: the behaviour policy is the notebook's `Beta` net (oracle.reinforce_oracle.beta_step, pinned on the notebook's own
    class by oracle/make_golden_beta.py), started from fx["g"]["beta_w0" / "beta_b0"] and TRAINED inside every step on the batch's
    action, as `_select_action_with_TopK_correction(state, beta_net.forward, action, ...)` does (recnn/nn/models.py:143-184 +
    notebook cell 3); returns its final parameters under "beta"."""
    from oracle import recnn_oracle as O
    from oracle import reinforce_oracle as R
    g = fx["g"]
    pol = {k: torch.from_numpy(g[f"policy.{k}"].copy()) for k in R.POLICY_ORDER}
    val = {k: torch.from_numpy(g[f"value.{k}"].copy()) for k in O.PARAM_ORDER}
    st = R.ReinforceState.create(pol, val, R.AdamDict(R.POLICY_ORDER, lr=fx["lr_p"], weight_decay=fx["wd_p"]),
                                 R.AdamDict(O.PARAM_ORDER, lr=fx["lr_v"], weight_decay=fx["wd_v"]), method=fx["method"], K=fx["K"])
    bs, beta = batches(fx), beta_fn(fx)
    if learned_beta:
        bp = {"w": torch.from_numpy(g["beta_w0"].copy()), "b": torch.from_numpy(g["beta_b0"].copy())}
        bopt = R.AdamDict(("w", "b"), lr=fx["lr_b"], weight_decay=fx["wd_b"])

        def beta(state, action=None, _bs=None):
            return R.beta_step(bp, bopt, state, action.argmax(1))[0]
    pi_draws, beta_draws = torch.from_numpy(g["pi_draws"]), torch.from_numpy(g["beta_draws"])
    losses = []
    for t in range(fx["steps"]):
        b = bs[t % 2]
        masks = [torch.from_numpy(m) for m in g["masks"][t]]
        basic = fx["method"] == "basic"
        scored = pi_draws[t] if (basic or fx["pi_source"] == "pi") else beta_draws[t]
        out = R.reinforce_step(st, b, scored, masks, step=t, beta_probs=None if basic else (beta(b["state"], b["action"]) if learned_beta else beta(b["state"])),
                               beta_action=None if basic else beta_draws[t])
        if out["policy"] is not None:
            losses.append([t, out["value"], out["policy"]])
    final = {"policy": st.policy, "value": st.value, "target_policy": st.target_policy, "target_value": st.target_value}
    if learned_beta:
        final["beta"] = bp
    return np.asarray(losses), final
