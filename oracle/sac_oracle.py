"""CPU oracle of one Soft Actor-Critic step -- TEST INFRASTRUCTURE ONLY (imported by tests/ and the golden generator, never by
recnn_amd).  Restates `soft_q_update` and the three networks of the reference's `examples/1. Vanilla RL/4. SAC.ipynb`
(code cells 5-8) on plain parameter dicts: the forward passes and losses are written out (no nn.Module, no nn.Dropout, no
torch.distributions); gradients come from autograd over those written-out formulas; the optimizer is the restated
torch.optim.Adam of oracle/recnn_oracle.py generalised to any key set (the notebook uses torch_optimizer.RAdam, a package that
is absent here and un-pinned there: Adam is the documented substitution, as for DDPG / TD3).

Pinned by tests/golden/sac_*.npz: runs of the notebook's OWN cells (exec'd from the .ipynb by oracle/make_golden_sac.py) with
torch.optim.Adam, logged dropout masks and z draws -- tests/test_sac_oracle.py replays them through this file.

Reference lines (SAC.ipynb code cells): StateCritic cell 5, SoftQ cell 6, StochasticActor.forward / evaluate cell 7,
soft_q_update cell 8, hyper-parameters cell 9.
"""
import math
from dataclasses import dataclass, field
from typing import Dict, Optional

import torch

Q_ORDER = ("w1", "b1", "w2", "b2", "w3", "b3")                       # StateCritic / SoftQ: nn.Module.parameters() order
POLICY_ORDER = ("w1", "b1", "w2", "b2", "wm", "bm", "ws", "bs")      # StochasticActor: linear1, linear2, mean_linear, log_std_linear


def critic_params_from_module(mod):
    return {"w1": mod.linear1.weight, "b1": mod.linear1.bias, "w2": mod.linear2.weight, "b2": mod.linear2.bias,
            "w3": mod.linear3.weight, "b3": mod.linear3.bias}


def policy_params_from_module(mod):
    return {"w1": mod.linear1.weight, "b1": mod.linear1.bias, "w2": mod.linear2.weight, "b2": mod.linear2.bias,
            "wm": mod.mean_linear.weight, "bm": mod.mean_linear.bias, "ws": mod.log_std_linear.weight, "bs": mod.log_std_linear.bias}


def snapshot(params):
    return {k: v.detach().cpu().float().clone() for k, v in params.items()}


def mlp3(p, x):
    """cells 5 / 6: relu(L1) -> relu(L2) -> L3."""
    h1 = torch.relu(x @ p["w1"].t() + p["b1"])
    h2 = torch.relu(h1 @ p["w2"].t() + p["b2"])
    return h2 @ p["w3"].t() + p["b3"]


def policy_forward(p, state, m1, m2, log_std_min, log_std_max):
    """cell 7 forward: dropout(0.5) = keep-mask * 2 after each relu; two heads; log_std clamped."""
    h1 = torch.relu(state @ p["w1"].t() + p["b1"])
    if m1 is not None:
        h1 = h1 * m1.float() * 2.0
    h2 = torch.relu(h1 @ p["w2"].t() + p["b2"])
    if m2 is not None:
        h2 = h2 * m2.float() * 2.0
    mean = h2 @ p["wm"].t() + p["bm"]
    log_std = torch.clamp(h2 @ p["ws"].t() + p["bs"], log_std_min, log_std_max)
    return mean, log_std


def policy_evaluate(p, state, z, m1, m2, log_std_min, log_std_max, epsilon=1e-6):
    """cell 7 evaluate: ONE scalar z; Normal(mean, std).log_prob at the SQUASHED action, minus log(1 - a^2 + eps); no sum."""
    mean, log_std = policy_forward(p, state, m1, m2, log_std_min, log_std_max)
    std = log_std.exp()
    action = torch.tanh(mean + z * std)
    log_prob = -((action - mean) ** 2) / (2 * std * std) - log_std - math.log(math.sqrt(2 * math.pi))
    log_prob = log_prob - torch.log(1 - action.pow(2) + epsilon)
    return action, log_prob, mean, log_std


@dataclass
class Adam:
    lr: float = 1e-3
    beta1: float = 0.9
    beta2: float = 0.999
    eps: float = 1e-8
    weight_decay: float = 0.0
    t: int = 0
    m: Dict[str, torch.Tensor] = field(default_factory=dict)
    v: Dict[str, torch.Tensor] = field(default_factory=dict)

    def step(self, p, g):
        """torch.optim.Adam (recnn_oracle.adam_step's formulas) over the keys of g."""
        self.t += 1
        bc1, bc2 = 1.0 - self.beta1 ** self.t, 1.0 - self.beta2 ** self.t
        for k, grad in g.items():
            if self.weight_decay:
                grad = grad + self.weight_decay * p[k]
            if k not in self.m:
                self.m[k], self.v[k] = torch.zeros_like(p[k]), torch.zeros_like(p[k])
            self.m[k] += (1.0 - self.beta1) * (grad - self.m[k])
            self.v[k].mul_(self.beta2).addcmul_(grad, grad, value=1.0 - self.beta2)
            p[k] -= (self.lr / bc1) * (self.m[k] / (self.v[k].sqrt() / math.sqrt(bc2) + self.eps))


@dataclass
class SACState:
    value: dict
    target_value: dict
    soft_q: dict
    policy: dict
    value_opt: Adam
    soft_q_opt: Adam
    policy_opt: Adam
    params: dict          # gamma, soft_tau, mean_lambda, std_lambda, z_lambda, log_std_min, log_std_max


def sac_step(st: SACState, batch, z, masks, step: int, learn: bool = True):
    """cell 8, in its order.  batch: state [B,S], action [B,A], reward [B], next_state [B,S], done [B]; z: python float / 0-dim
    tensor; masks: (m1, m2) keep-masks of the policy's two dropouts (None: eval)."""
    P = st.params
    state, action, next_state = batch["state"], batch["action"], batch["next_state"]
    reward, done = batch["reward"].reshape(-1, 1), batch["done"].reshape(-1, 1)
    z = torch.as_tensor(z, dtype=torch.float32)
    m1, m2 = masks if masks is not None else (None, None)

    def leaf(p):
        return {k: v.detach().clone().requires_grad_(True) for k, v in p.items()}

    # ---- soft Q
    ql = leaf(st.soft_q)
    expected_softq = mlp3(ql, torch.cat([state, action], 1))
    vl = leaf(st.value)
    expected_value = mlp3(vl, state)
    pl = leaf(st.policy)
    next_action, log_prob, mean, log_std = policy_evaluate(pl, state, z, m1, m2, P["log_std_min"], P["log_std_max"])
    with torch.no_grad():
        target_value = mlp3(st.target_value, next_state)
        next_q = reward + (1 - done) * P["gamma"] * target_value
    q_loss = ((expected_softq - next_q) ** 2).mean()
    if learn:
        g = torch.autograd.grad(q_loss, [ql[k] for k in Q_ORDER])
        with torch.no_grad():
            st.soft_q_opt.step(st.soft_q, dict(zip(Q_ORDER, g)))
    # ---- state value (Q with the UPDATED weights)
    with torch.no_grad():
        expected_next_softq = mlp3(st.soft_q, torch.cat([state, next_action.detach()], 1))
        next_value = expected_next_softq - log_prob.detach()                  # [B,1] - [B,A]
    value_loss = ((expected_value - next_value) ** 2).mean()
    if learn:
        g = torch.autograd.grad(value_loss, [vl[k] for k in Q_ORDER])
        with torch.no_grad():
            st.value_opt.step(st.value, dict(zip(Q_ORDER, g)))
            for k in Q_ORDER:                                                 # soft_update: target*(1-tau) + net*tau
                st.target_value[k].copy_(st.target_value[k] * (1.0 - P["soft_tau"]) + st.value[k] * P["soft_tau"])
    # ---- policy
    log_prob_target = (expected_next_softq - expected_value).detach()
    policy_loss = (log_prob * (log_prob - log_prob_target).detach()).mean()
    policy_loss = policy_loss + P["mean_lambda"] * mean.pow(2).mean() + P["std_lambda"] * log_std.pow(2).mean() \
        + P["z_lambda"] * z.pow(2)
    if learn:
        g = torch.autograd.grad(policy_loss, [pl[k] for k in POLICY_ORDER])
        with torch.no_grad():
            st.policy_opt.step(st.policy, dict(zip(POLICY_ORDER, g)))
    return {"value": float(value_loss.detach()), "softq": float(q_loss.detach()), "policy": float(policy_loss.detach()), "step": step,
            "next_action": next_action.detach(), "log_prob": log_prob.detach()}
