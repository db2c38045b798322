"""Replays `tests/golden/bcq_small.npz` (a run of the REAL reference's bcq_update, oracle/make_golden_bcq.py) through either
the CPU oracle or the HIP product path; shared by the CPU and GPU tests."""
import numpy as np
import torch


def load(path):
    g = np.load(path)
    S, A, L, H, B, n, steps, seed, pstep, sample = (int(x) for x in g["dims"])
    lr_g, lr_v, lr_p, wd_v, gamma, tau = (float(x) for x in g["hyper"])
    return dict(g=g, S=S, A=A, L=L, H=H, B=B, n=n, steps=steps, seed=seed, pstep=pstep, sample=sample, lr_g=lr_g, lr_v=lr_v,
                lr_p=lr_p, wd_v=wd_v, gamma=gamma, tau=tau)


def build_nets(fx):
    """The construction order of oracle/make_golden_bcq.py::build_nets with recnn_amd's modules: equal seeds must give the
    reference's initial weights (checked against the fixture's checksums by `check_init`)."""
    from recnn_amd.nn import models as M
    torch.manual_seed(fx["seed"])
    S, A, L, H = fx["S"], fx["A"], fx["L"], fx["H"]
    gen = M.bcqGenerator(S, A, L)
    pert = M.bcqPerturbator(S, A, H)
    tpert = M.bcqPerturbator(S, A, H)
    v1, v2 = M.Critic(S, A, H, 2e-1), M.Critic(S, A, H, 2e-1)
    tv1, tv2 = M.Critic(S, A, H), M.Critic(S, A, H)
    for t in (tpert, tv1, tv2):
        t.eval()
    return gen, pert, tpert, v1, v2, tv1, tv2


def check_init(fx, gen, pert, v1):
    from oracle import recnn_oracle as O
    from oracle import bcq_oracle as Q
    sums = {k: float(v.double().abs().sum()) for k, v in Q.generator_params_from_module(gen).items()}
    sums.update({"pert." + k: float(v.double().abs().sum()) for k, v in O.params_from_module(pert).items()})
    sums.update({"v1." + k: float(v.double().abs().sum()) for k, v in O.params_from_module(v1).items()})
    for k, ref in zip(fx["g"]["init_keys"], fx["g"]["init_abs"]):
        assert abs(sums[str(k)] - float(ref)) <= 1e-9 * abs(float(ref)), (k, sums[str(k)], float(ref))


def batches(fx, device=None):
    out = []
    for i in range(2):
        b = {k: torch.from_numpy(fx["g"][f"batch{i}.{k}"].copy()) for k in ("state", "action", "reward", "next_state", "done")}
        out.append({k: v.to(device) for k, v in b.items()} if device is not None else b)
    return out


def params_of(fx):
    return {"gamma": fx["gamma"], "soft_tau": fx["tau"], "n_generator_samples": fx["n"], "perturbator_step": fx["pstep"]}


def replay_oracle(fx):
    from oracle import recnn_oracle as O
    from oracle import bcq_oracle as Q
    from oracle.reinforce_oracle import AdamDict
    gen, pert, tpert, v1, v2, tv1, tv2 = build_nets(fx)
    check_init(fx, gen, pert, v1)
    P = O.params_from_module
    # algo-style hard sync of the targets (soft_update with tau = 1)
    st = Q.BCQState(Q.generator_params_from_module(gen), P(pert), P(pert), P(v1), P(v1), P(v2), P(v2),
                    AdamDict(Q.GEN_ORDER, lr=fx["lr_g"]), AdamDict(O.PARAM_ORDER, lr=fx["lr_p"]),
                    AdamDict(O.PARAM_ORDER, lr=fx["lr_v"], weight_decay=fx["wd_v"]), params=params_of(fx))
    g, bs = fx["g"], batches(fx)
    losses = []
    for t in range(fx["steps"]):
        out = Q.bcq_step(st, bs[t % 2], torch.from_numpy(g["eps"][t]), torch.from_numpy(g["z_next"][t]),
                         torch.from_numpy(g["z_cur"][t]), [torch.from_numpy(m) for m in g["masks"][t]], step=t)
        losses.append([out["value"], out["perturbator"], out["generator"]])
    final = {"generator": st.generator, "perturbator": st.perturbator, "target_perturbator": st.target_perturbator,
             "value1": st.value1, "target_value1": st.target_value1, "value2": st.value2, "target_value2": st.target_value2}
    return np.asarray(losses), final


def compare_final(fx, final, rtol, atol_frac=1e-6):
    """`final`: {net: {tensor name: tensor}} against the strided samples and absolute sums of the fixture.  Returns the
    worst element-wise error relative to the tensor's largest magnitude."""
    g, worst = fx["g"], 0.0
    for key in g.files:
        if not key.startswith("final."):
            continue
        _, net, name = key.split(".", 2)
        got = final[net][name].detach().cpu().double().reshape(-1)[::fx["sample"]].numpy()
        ref = g[key].astype(np.float64)
        scale = max(float(np.abs(ref).max()), 1e-30)
        err = float(np.abs(got - ref).max() / scale)
        worst = max(worst, err)
        assert err < rtol, (net, name, err)
        s_got = float(final[net][name].detach().cpu().double().abs().sum())
        s_ref = float(g[f"final_abs.{net}.{name}"])
        assert abs(s_got - s_ref) <= rtol * abs(s_ref) + atol_frac, (net, name, s_got, s_ref)
    return worst
