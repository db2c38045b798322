"""Replays a `tests/golden/sac_*.npz` fixture (a run of the reference notebook's own SAC cells, oracle/make_golden_sac.py) through
the CPU oracle; shared by the CPU and GPU tests."""
import numpy as np
import torch

PARAMS = {"gamma": 0.99, "soft_tau": 0.001, "mean_lambda": 1e-3, "std_lambda": 1e-3, "z_lambda": 1e-10}
ACTOR = {"mean_initw": 1e-1, "std_initw": 6e-1, "log_std_min": -2, "log_std_max": 2}


def load(path):
    g = np.load(path)
    S, A, H, B, steps, seed = (int(x) for x in g["dims"])
    lr, wd = (float(x) for x in g["hyper"])
    return dict(g=g, S=S, A=A, H=H, B=B, steps=steps, lr=lr, wd=wd)


def batches(fx, device=None):
    out = []
    for i in range(2):
        b = {k: torch.from_numpy(fx["g"][f"batch{i}.{k}"].copy()) for k in ("state", "action", "reward", "next_state", "done")}
        out.append({k: v.to(device) for k, v in b.items()} if device is not None else b)
    return out


def replay_oracle(fx):
    from oracle import sac_oracle as SO
    g = fx["g"]
    take = lambda tag, order: {k: torch.from_numpy(g[f"{tag}.{k}"].copy()) for k in order}
    st = SO.SACState(value=take("value", SO.Q_ORDER), target_value=take("value", SO.Q_ORDER), soft_q=take("soft_q", SO.Q_ORDER),
                     policy=take("policy", SO.POLICY_ORDER), value_opt=SO.Adam(lr=fx["lr"], weight_decay=fx["wd"]),
                     soft_q_opt=SO.Adam(lr=fx["lr"], weight_decay=fx["wd"]), policy_opt=SO.Adam(lr=fx["lr"], weight_decay=fx["wd"]),
                     params=dict(PARAMS, **ACTOR))
    bs = batches(fx)
    losses = []
    for t in range(fx["steps"]):
        masks = [torch.from_numpy(m) for m in g["masks"][t]]
        out = SO.sac_step(st, bs[t % 2], float(g["z"][t]), masks, step=t)
        losses.append([t, out["value"], out["softq"], out["policy"]])
    return np.asarray(losses), {"value": st.value, "target_value": st.target_value, "soft_q": st.soft_q, "policy": st.policy}
