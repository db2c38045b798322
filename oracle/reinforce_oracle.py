"""CPU oracle for the REINFORCE step of RecNN.

TEST INFRASTRUCTURE -- NOT A PRODUCT PATH.  Only `tests/` may import this module; `recnn_amd` never does.

A restatement (torch-CPU fp32, backward written out by hand) of

  recnn/nn/models.py:76-184        DiscreteActor (forward, Categorical sample / log_prob, correction, lambda_K)
  recnn/nn/update/reinforce.py:10-66   ChooseREINFORCE (three estimators, discounted + normalised returns)
  recnn/nn/update/reinforce.py:69-129  reinforce_update
  recnn/nn/update/misc.py:10-55    value_update (critic over [state | action distribution])
  recnn/data/utils.py:84-120       batch_contstate_discaction (one-hot action rows)
  torch/distributions/categorical.py   Categorical(probs): probs / probs.sum(-1), clamp to [eps, 1-eps], log, gather
                                       (torch 2.x, the version of this image; third-party to the reference)

Sampled actions are INPUTS here (the reference draws them with torch.multinomial from the global generator; the HIP path
has its own counter-based sampler -- equality of the two is in distribution only, so parity is checked with injected
actions: `oracle/make_golden_reinforce.py` patches `Categorical.sample` in the real reference the same way).

Pinning: checked against the real reference (imported from /root/reference in the build container) by
`oracle/make_golden_reinforce.py`, which writes `tests/golden/reinforce_*.npz`; `tests/test_oracle_golden.py` re-checks the
oracle against those fixtures.  Optimizer of record: torch.optim.Adam injected through `algo.optimizers[...]`.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import math
import torch

from . import recnn_oracle as O

EPS = torch.finfo(torch.float32).eps
POLICY_ORDER = ("w1", "b1", "w2", "b2")   # == DiscreteActor.parameters() order


def policy_params_from_module(mod) -> Dict[str, torch.Tensor]:
    return {"w1": mod.linear1.weight.detach().cpu().float().clone(), "b1": mod.linear1.bias.detach().cpu().float().clone(),
            "w2": mod.linear2.weight.detach().cpu().float().clone(), "b2": mod.linear2.bias.detach().cpu().float().clone()}


def policy_forward(p, state: torch.Tensor):
    """softmax(L2(relu(L1(state))))  (models.py:95-99; F.softmax without dim = dim 1 for a 2-D input)."""
    h = torch.relu(torch.addmm(p["b1"], state, p["w1"].t()))
    logits = torch.addmm(p["b2"], h, p["w2"].t())
    probs = torch.softmax(logits, dim=1)
    return probs, (state, h)


def categorical_log_prob(probs: torch.Tensor, actions: torch.Tensor):
    """Categorical(probs).log_prob(actions): normalise, clamp, log, gather.  Returns (log_prob, clamped flag)."""
    q = probs / probs.sum(-1, keepdim=True)
    qa = q.gather(1, actions.view(-1, 1)).squeeze(1)
    clamped = (qa < EPS) | (qa > 1 - EPS)
    return torch.log(qa.clamp(min=EPS, max=1 - EPS)), clamped


def policy_backward(p, cache, probs, actions, g_lp, clamped):
    """Gradient of sum_b g_lp[b] * log_prob[b] w.r.t. the policy parameters.
    d log_prob / d logits = onehot(a) - p / sum(p) (the normalisation cancels), zero where the clamp was active."""
    state, h = cache
    g = torch.where(clamped, torch.zeros_like(g_lp), g_lp)
    dlogits = -(probs / probs.sum(-1, keepdim=True)) * g.view(-1, 1)
    dlogits[torch.arange(probs.shape[0]), actions] += g
    grads = {"w2": dlogits.t() @ h, "b2": dlogits.sum(0)}
    dz1 = (dlogits @ p["w2"]) * (h > 0).to(dlogits.dtype)
    grads["w1"] = dz1.t() @ state
    grads["b1"] = dz1.sum(0)
    return grads


def discounted_returns(rewards: Sequence[torch.Tensor], gamma: float = 0.99, eps: float = 0.0001) -> torch.Tensor:
    """reinforce.py:45-53: R = r + 0.99 R backwards, then (R - mean) / (std + 1e-4), unbiased std, fp32."""
    run = torch.zeros((), dtype=torch.float32)
    out = []
    for r in reversed(list(rewards)):
        run = torch.as_tensor(r, dtype=torch.float32) + gamma * run
        out.insert(0, run)
    ret = torch.stack(out)
    return (ret - ret.mean()) / (ret.std() + eps)


def reinforce_loss(method: str, lps, blps, returns, K: int):
    """Policy loss of one episode and d loss / d log_prob per step (reinforce.py:16-43).
      basic: sum -lp R                                   -> -R
      corr : sum c (-lp) R,  c = e^lp / e^blp            -> -R c (1 + lp)         (c is NOT detached in the reference)
      topk : sum l c (-lp) R, l = K (1 - e^lp)^(K-1)     -> -R c (l (1 + lp) + lp dl),  dl = -K (K-1) (1 - e^lp)^(K-2) e^lp
    """
    loss = torch.zeros((), dtype=torch.float32)
    glps = []
    for t, lp in enumerate(lps):
        R = returns[t]
        if method == "basic":
            loss = loss + (-lp * R).sum()
            glps.append(torch.full_like(lp, -1.0) * R)
            continue
        c = torch.exp(lp) / torch.exp(blps[t])
        if method == "corr":
            loss = loss + (c * -lp * R).sum()
            glps.append(-R * c * (1.0 + lp))
        elif method == "topk":
            pi = torch.exp(lp)
            l = K * (1 - pi) ** (K - 1)
            dl = -K * (K - 1) * (1 - pi) ** (K - 2) * pi
            loss = loss + (l * c * -lp * R).sum()
            glps.append(-R * c * (l * (1.0 + lp) + lp * dl))
        else:
            raise ValueError(method)
    return loss, glps


@dataclass
class AdamDict:
    """torch.optim.Adam over an ordered dict of tensors (same arithmetic as recnn_oracle.adam_step)."""
    order: Sequence[str]
    lr: float = 1e-3
    beta1: float = 0.9
    beta2: float = 0.999
    eps: float = 1e-8
    weight_decay: float = 0.0
    t: int = 0
    m: Dict[str, torch.Tensor] = field(default_factory=dict)
    v: Dict[str, torch.Tensor] = field(default_factory=dict)

    def step(self, p, g):
        self.t += 1
        bc1 = 1.0 - self.beta1 ** self.t
        bc2_sqrt = math.sqrt(1.0 - self.beta2 ** self.t)
        for k in self.order:
            grad = g[k]
            if self.weight_decay != 0.0:
                grad = grad + self.weight_decay * p[k]
            if k not in self.m:
                self.m[k] = torch.zeros_like(p[k])
                self.v[k] = torch.zeros_like(p[k])
            self.m[k] += (1.0 - self.beta1) * (grad - self.m[k])
            self.v[k].mul_(self.beta2).addcmul_(grad, grad, value=1.0 - self.beta2)
            p[k] -= (self.lr / bc1) * (self.m[k] / (self.v[k].sqrt() / bc2_sqrt + self.eps))


def soft_update(net, target, tau, order):
    for k in order:
        target[k] = target[k] * (1.0 - tau) + net[k] * tau


@dataclass
class ReinforceState:
    policy: Dict[str, torch.Tensor]
    value: Dict[str, torch.Tensor]
    target_policy: Dict[str, torch.Tensor]
    target_value: Dict[str, torch.Tensor]
    policy_opt: AdamDict
    value_opt: AdamDict
    method: str = "basic"
    K: int = 10
    gamma: float = 0.99
    min_value: float = -10.0
    max_value: float = 10.0
    policy_step: int = 10
    soft_tau: float = 0.001
    rewards: List[torch.Tensor] = field(default_factory=list)
    episode: List[dict] = field(default_factory=list)

    @staticmethod
    def create(policy, value, policy_opt, value_opt, **kw):
        return ReinforceState(policy, value, O.clone_params(policy), O.clone_params(value), policy_opt, value_opt, **kw)


def reinforce_step(st: ReinforceState, batch, pi_action: torch.Tensor, masks: Sequence[Optional[torch.Tensor]], step: int,
                   beta_probs: Optional[torch.Tensor] = None, beta_action: Optional[torch.Tensor] = None):
    """One `reinforce_update` call (reinforce.py:69-129).
    batch: state, action (one-hot [B, N]), reward [B], next_state, done [B].
    pi_action: the action the policy is scored on (for action_source pi = "beta" pass the behaviour policy's action).
    beta_probs / beta_action: the behaviour policy's probabilities and drawn action (correction estimators only).
    masks: 4 keep-masks of the learning critic in call order (2 for the reward forward, 2 for the TD forward), or Nones.
    Returns {"value": loss, "policy": loss or None, "log_prob": ..., "reward": ...}.
    """
    state, action, reward, next_state, done = (torch.as_tensor(batch[k]).float() for k in
                                               ("state", "action", "reward", "next_state", "done"))
    reward, done = reward.view(-1, 1), done.view(-1, 1)
    # act (models.py:103-111 / :113-141)
    probs, cache = policy_forward(st.policy, state)
    lp, clamped = categorical_log_prob(probs, pi_action)
    blp = None
    if beta_probs is not None:
        blp, _ = categorical_log_prob(torch.as_tensor(beta_probs).float(), beta_action)
    st.episode.append({"cache": cache, "probs": probs, "action": pi_action, "lp": lp, "blp": blp, "clamped": clamped})
    # reward = mean critic score of the action distribution (reinforce.py:96-97)
    q, _ = O.critic_forward(st.value, state, probs, masks[0], masks[1])
    st.rewards.append(q.mean())
    # critic TD update (misc.py:10-55)
    nprobs, _ = policy_forward(st.target_policy, next_state)
    tq, _ = O.critic_forward(st.target_value, next_state, nprobs)
    expected = torch.clamp(O.temporal_difference(reward, done, st.gamma, tq), st.min_value, st.max_value)
    v, vcache = O.critic_forward(st.value, state, action, masks[2], masks[3])
    value_loss = torch.pow(v - expected, 2).mean()
    dv = 2.0 * (v - expected) / v.numel()
    gv, _, _ = O.mlp_backward(st.value, vcache, dv, train=masks[2] is not None)
    st.value_opt.step(st.value, gv)
    out = {"value": float(value_loss), "policy": None, "log_prob": lp, "reward": float(st.rewards[-1])}
    if step % st.policy_step == 0 and step > 0:
        returns = discounted_returns(st.rewards)
        loss, glps = reinforce_loss(st.method, [e["lp"] for e in st.episode], [e["blp"] for e in st.episode], returns, st.K)
        grads = None
        for e, g in zip(st.episode, glps):
            ge = policy_backward(st.policy, e["cache"], e["probs"], e["action"], g, e["clamped"])
            grads = ge if grads is None else {k: grads[k] + ge[k] for k in POLICY_ORDER}
        st.policy_opt.step(st.policy, grads)
        st.rewards, st.episode = [], []
        soft_update(st.value, st.target_value, st.soft_tau, O.PARAM_ORDER)
        soft_update(st.policy, st.target_policy, st.soft_tau, POLICY_ORDER)
        out["policy"] = float(loss)
    return out


# --------------------------------------------------------------------------------------
# The learned behaviour policy `Beta` of the Top-K correction notebook
# (examples/2. REINFORCE TopK Off Policy Correction/3. TopK Reinforce Off Policy Correction.ipynb, cell 3; consumed by
# recnn/nn/models.py:113-141,143-184 as `beta(state, action=...) -> probabilities`).  Pinned by oracle/make_golden_beta.py, which
# exec()s the notebook's own class.
# --------------------------------------------------------------------------------------


def beta_step(p: Dict[str, torch.Tensor], opt: "AdamDict", state: torch.Tensor, target: torch.Tensor):
    """One `Beta.forward(state, action)` of the notebook with torch.optim.Adam in place of the absent torch_optimizer.RAdam:

        probs = Softmax()(Linear(state))                       (returned: the probabilities BEFORE this call's optimizer step)
        loss  = CrossEntropyLoss()(probs, action.argmax(1))    -- the cross entropy of the PROBABILITIES taken as logits (sic)
        zero_grad(); loss.backward(); optim.step()

    Backward by hand: d loss / d probs = (softmax(probs) - onehot(target)) / B;  through the softmax:
    d logits = probs * (g - sum_j g_j probs_j);  dW = d logits^T state, db = column sums.  p = {"w": [N, K], "b": [N]}, updated
    in place.  Returns (probs, loss)."""
    B = state.shape[0]
    logits = torch.addmm(p["b"], state, p["w"].t())
    probs = torch.softmax(logits, dim=1)
    q = torch.softmax(probs, dim=1)                       # log_softmax of the "logits" CrossEntropyLoss is given
    loss = -torch.log(q.gather(1, target.view(-1, 1)).squeeze(1)).mean()
    g = q.clone()
    g[torch.arange(B), target] -= 1.0
    g /= B
    dlogits = probs * (g - (g * probs).sum(1, keepdim=True))
    grads = {"w": dlogits.t() @ state, "b": dlogits.sum(0)}
    opt.step(p, grads)
    return probs, float(loss)
