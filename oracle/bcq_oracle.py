"""CPU oracle for the BCQ step of RecNN.

TEST INFRASTRUCTURE -- NOT A PRODUCT PATH.  Only `tests/` may import this module; `recnn_amd` never does.

A restatement (torch-CPU fp32, backward written out by hand) of

  recnn/nn/models.py:216-242    bcqPerturbator  (Critic-shaped MLP over [state | action], output + action)
  recnn/nn/models.py:245-295    bcqGenerator    (VAE: e1, e2 -> mean / clamp(log_std, -4, 15); z = mean + std * eps;
                                                 decode(state, z): d1, d2, d3; decode without z draws clamp(N(0,1), +-0.5))
  recnn/nn/update/bcq.py:11-179 bcq_update      (generator step, critic step on the max over n_generator_samples perturbed
                                                 candidate actions, perturbator step every `perturbator_step`, soft updates)

Quirks of the reference kept on purpose:
  * bcq.py:2 imports `torch.functional as F`, which has no `mse_loss`: as written `bcq_update` raises AttributeError on its
    first line of arithmetic.  The oracle of record is the reference with that one name re-pointed at
    `torch.nn.functional` (oracle/make_golden_bcq.py does exactly that and nothing else to the update function);
  * bcq.py:105-106 read `target_value_net1` for BOTH target Q values, so 0.75 min + 0.25 max collapses to
    0.75 q + 0.25 q of one critic; `value_net2` never receives a gradient (its optimizer steps over `grad is None`
    parameters, a no-op) yet `target_value_net2` is still soft-updated towards it;
  * the perturbator loss is evaluated (and returned) every step, its backward only on `step % perturbator_step == 0`,
    followed by the `clip_grad_norm_(.., -1, 1)` L1-normalise-and-flip quirk (bcq.py:136).

Normal draws and dropout keep-masks are INPUTS (the reference draws them from the global CPU generator in a fixed order:
eps [B, L], z_next [B n, L], 2 masks (critic), z_cur [B, L], 2 masks (perturbator), 2 masks (critic)).

Pinning: checked against the real reference (imported from /root/reference in the build container) by
`oracle/make_golden_bcq.py`, which writes `tests/golden/bcq_small.npz`; `tests/test_oracle_golden.py` re-checks the oracle
against that fixture.  Optimizer of record: torch.optim.Adam passed in the `optimizer` dict.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, Optional, Sequence

import torch

from . import recnn_oracle as O
from .reinforce_oracle import AdamDict

GEN_ORDER = ("e1.w", "e1.b", "e2.w", "e2.b", "mean.w", "mean.b", "log_std.w", "log_std.b",
             "d1.w", "d1.b", "d2.w", "d2.b", "d3.w", "d3.b")            # == bcqGenerator.parameters() order
LOG_STD_MIN, LOG_STD_MAX = -4.0, 15.0
Z_CLIP = 0.5


def generator_params_from_module(mod) -> Dict[str, torch.Tensor]:
    out = {}
    for name in ("e1", "e2", "mean", "log_std", "d1", "d2", "d3"):
        lin = getattr(mod, name)
        out[name + ".w"] = lin.weight.detach().cpu().float().clone()
        out[name + ".b"] = lin.bias.detach().cpu().float().clone()
    return out


def _encoder_as_mlp(g):
    return {"w1": g["e1.w"], "b1": g["e1.b"], "w2": g["e2.w"], "b2": g["e2.b"],
            "w3": torch.cat([g["mean.w"], g["log_std.w"]], 0), "b3": torch.cat([g["mean.b"], g["log_std.b"]], 0)}


def _decoder_as_mlp(g):
    return {"w1": g["d1.w"], "b1": g["d1.b"], "w2": g["d2.w"], "b2": g["d2.b"], "w3": g["d3.w"], "b3": g["d3.b"]}


def decode(g, state, z):
    """bcqGenerator.decode with an explicit latent (models.py:284-295): d3(relu(d2(relu(d1([state | z])))))."""
    return O.mlp_forward(_decoder_as_mlp(g), torch.cat([state, z], 1))


def generator_forward(g, state, action, eps):
    """bcqGenerator.forward (models.py:265-282).  Returns (u, mean, std, caches)."""
    L = g["mean.w"].shape[0]
    ml, enc_cache = O.mlp_forward(_encoder_as_mlp(g), torch.cat([state, action], 1))
    mean, raw = ml[:, :L], ml[:, L:]
    std = torch.exp(raw.clamp(LOG_STD_MIN, LOG_STD_MAX))
    z = mean + std * eps
    u, dec_cache = decode(g, state, z)
    return u, mean, std, (enc_cache, dec_cache, raw)


def generator_loss_and_grads(g, state, action, eps):
    """recon = mse(u, action); KL = -0.5 mean(1 + log(std^2) - mean^2 - std^2); loss = recon + 0.5 KL (bcq.py:78-81)
    and its gradient w.r.t. every generator tensor."""
    B, A = action.shape
    S = state.shape[1]
    L = g["mean.w"].shape[0]
    u, mean, std, (enc_cache, dec_cache, raw) = generator_forward(g, state, action, eps)
    recon = ((u - action) ** 2).mean()
    kl = -0.5 * (1 + torch.log(std.pow(2)) - mean.pow(2) - std.pow(2)).mean()
    loss = recon + 0.5 * kl
    du = 2.0 * (u - action) / (B * A)
    gd, dx, _ = O.mlp_backward(_decoder_as_mlp(g), dec_cache, du, train=False, need_dx=True)
    dz = dx[:, S:]
    c = 0.5 * (-0.5) / (B * L)                                  # d loss / d (the KL bracket), per element
    dmean = dz + c * (-2.0 * mean)
    dstd = dz * eps + c * (2.0 / std - 2.0 * std)
    inside = ((raw >= LOG_STD_MIN) & (raw <= LOG_STD_MAX)).to(std.dtype)      # clamp passes the gradient on [min, max]
    draw = dstd * std * inside
    ge, _, _ = O.mlp_backward(_encoder_as_mlp(g), enc_cache, torch.cat([dmean, draw], 1), train=False)
    grads = {"e1.w": ge["w1"], "e1.b": ge["b1"], "e2.w": ge["w2"], "e2.b": ge["b2"],
             "mean.w": ge["w3"][:L], "mean.b": ge["b3"][:L], "log_std.w": ge["w3"][L:], "log_std.b": ge["b3"][L:],
             "d1.w": gd["w1"], "d1.b": gd["b1"], "d2.w": gd["w2"], "d2.b": gd["b2"], "d3.w": gd["w3"], "d3.b": gd["b3"]}
    return float(loss), grads, {"recon": u, "mean": mean, "std": std, "recon_loss": float(recon), "kl_loss": float(kl)}


def perturbator_forward(p, state, action, m1=None, m2=None):
    """bcqPerturbator.forward (models.py:234-242): MLP([state | action]) + action."""
    out, cache = O.mlp_forward(p, torch.cat([state, action], 1), m1, m2)
    return out + action, cache


@dataclass
class BCQState:
    generator: Dict[str, torch.Tensor]
    perturbator: Dict[str, torch.Tensor]
    target_perturbator: Dict[str, torch.Tensor]
    value1: Dict[str, torch.Tensor]
    target_value1: Dict[str, torch.Tensor]
    value2: Dict[str, torch.Tensor]
    target_value2: Dict[str, torch.Tensor]
    generator_opt: AdamDict
    perturbator_opt: AdamDict
    value_opt: AdamDict
    params: Dict[str, float] = field(default_factory=lambda: {"gamma": 0.99, "soft_tau": 0.001, "n_generator_samples": 10,
                                                              "perturbator_step": 30})


def bcq_step(st: BCQState, batch, eps, z_next, z_cur, masks: Sequence[Optional[torch.Tensor]], step: int):
    """One learn=True call of bcq_update (bcq.py:69-179).  masks: 6 keep-masks [B, H] in consumption order.
    z_next / z_cur are the RAW N(0,1) draws of decode(); the clamp to +-0.5 happens here (models.py:287-288)."""
    f = lambda x: torch.as_tensor(x, dtype=torch.float32)
    state, action, next_state = f(batch["state"]), f(batch["action"]), f(batch["next_state"])
    reward, done = f(batch["reward"]).view(-1, 1), f(batch["done"]).view(-1, 1)
    B, S = state.shape
    n = int(st.params["n_generator_samples"])
    # generator (VAE) step
    gen_loss, gg, _ = generator_loss_and_grads(st.generator, state, action, f(eps))
    st.generator_opt.step(st.generator, gg)
    # critic step on the best of n perturbed candidates per row
    state_rep = torch.repeat_interleave(next_state, n, 0)
    sampled, _ = decode(st.generator, state_rep, f(z_next).clamp(-Z_CLIP, Z_CLIP))
    perturbed, _ = perturbator_forward(st.target_perturbator, state_rep, sampled)
    q1, _ = O.critic_forward(st.target_value1, state_rep, perturbed)
    q2, _ = O.critic_forward(st.target_value1, state_rep, perturbed)          # bcq.py:106: net1 again
    target = 0.75 * torch.min(q1, q2)
    target = target + 0.25 * torch.max(q1, q2)
    target = target.view(B, -1).max(1)[0].view(-1, 1)
    expected = reward + (1.0 - done) * st.params["gamma"] * target
    q, cache = O.critic_forward(st.value1, state, action, masks[0], masks[1])
    value_loss = float(((q - expected) ** 2).mean())
    gv, _, _ = O.mlp_backward(st.value1, cache, 2.0 * (q - expected) / B)
    st.value_opt.step(st.value1, gv)
    # perturbator
    sampled, _ = decode(st.generator, state, f(z_cur).clamp(-Z_CLIP, Z_CLIP))
    perturbed, pcache = perturbator_forward(st.perturbator, state, sampled, masks[2], masks[3])
    qp, ccache = O.critic_forward(st.value1, state, perturbed, masks[4], masks[5])
    pert_loss = float((-qp).mean())
    if step % int(st.params["perturbator_step"]) == 0:
        _, dxa, _ = O.mlp_backward(st.value1, ccache, torch.full_like(qp, -1.0 / B), need_dx=True, need_dw=False)
        gp, _, _ = O.mlp_backward(st.perturbator, pcache, dxa[:, S:])
        coef = O.clip_grad_quirk_scale(gp)
        st.perturbator_opt.step(st.perturbator, {k: v * coef for k, v in gp.items()})
    tau = st.params["soft_tau"]
    O.soft_update(st.value1, st.target_value1, tau)
    O.soft_update(st.value2, st.target_value2, tau)
    O.soft_update(st.perturbator, st.target_perturbator, tau)
    return {"value": value_loss, "perturbator": pert_loss, "generator": gen_loss, "step": step,
            "expected": expected, "target_value": target}
