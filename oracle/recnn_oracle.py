"""CPU oracle for the RecNN DDPG/TD3 inner training step.

TEST INFRASTRUCTURE -- NOT A PRODUCT PATH.  Only `tests/`, `__graft_entry__.smoke()` and
the `cpu_baseline` leg of `bench.py` may import this module.  `recnn_amd` never does; the
product path raises when the HIP library is missing.

This file is a *restatement* (numpy for the integer / copy work, torch-CPU fp32 for the
floating-point contractions) of the reference algorithm, written from the behaviour of
the reference files below.  Every function cites the reference lines it follows
(paths relative to /root/reference):

  recnn/data/utils.py:7-10      rolling_window
  recnn/data/utils.py:161-187   prepare_batch_static_size
  recnn/data/utils.py:51-81     batch_tensor_embeddings
  recnn/data/utils.py:265-276   get_base_batch
  recnn/nn/models.py:41-73      Actor
  recnn/nn/models.py:187-213    Critic
  recnn/nn/update/misc.py:6-55  temporal_difference, value_update
  recnn/nn/update/ddpg.py:58-104  ddpg_update
  recnn/nn/update/td3.py:66-150   td3_update
  recnn/utils/misc.py:1-5       soft_update
  recnn/nn/algo.py:65-179       DDPG / TD3 (targets = deepcopy + hard sync, params dicts)

Pinning status: the restatement is checked against the REAL reference, imported from
/root/reference in the build container by `oracle/make_golden.py` (which also writes the
fixtures under `tests/golden/`), and against those committed fixtures by
`tests/test_oracle_golden.py`.  The reference's own tests hold no golden vectors
(`.circleci/tests/learning.py` asserts only signs), so the fixtures are the pin.
The optimizer of record is `torch.optim.Adam` injected through the reference's public
`optimizers[...]` dict; the reference default `torch_optimizer.Ranger` is an absent,
un-pinned third-party package -> **parity unpinned** at that boundary.

The backward pass is written out by hand (no autograd) on purpose: it is the
specification the HIP kernels implement, and the intermediate tensors it returns let the
GPU tests check kernels one at a time.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

# --------------------------------------------------------------------------------------
# 1. Sampler + embedding gather (integer / copy work: numpy, bit-exact)
# --------------------------------------------------------------------------------------


def rolling_window(a: np.ndarray, window: int) -> np.ndarray:
    """All length-`window` sliding windows of a 1-D array (utils.py:7-10).

    The reference builds an `as_strided` view; the values are
    out[t, j] = a[t + j] for t in [0, L-window], j in [0, window).
    """
    n = a.shape[-1] - window + 1
    if n <= 0:
        return np.empty((0, window), dtype=a.dtype)
    idx = np.arange(n)[:, None] + np.arange(window)[None, :]
    return a[idx]


def frame_windows(user_items: Sequence[np.ndarray], user_ratings: Sequence[np.ndarray], frame_size: int):
    """utils.py:161-187 up to the `embed_batch` call.

    Returns items int64[B, F+1], ratings float32[B, F+1] (float64 -> float32 cast as in
    `torch.tensor(ratings_t).float()`, utils.py:178) and sizes int64[U].
    """
    items = np.concatenate([rolling_window(np.asarray(i), frame_size + 1) for i in user_items], 0)
    ratings = np.concatenate([rolling_window(np.asarray(r), frame_size + 1) for r in user_ratings], 0)
    sizes = np.asarray([len(i) for i in user_items], dtype=np.int64)
    return items.astype(np.int64), ratings.astype(np.float32), sizes


def embed_frames(items: np.ndarray, ratings: np.ndarray, sizes: np.ndarray, table: np.ndarray, frame_size: int):
    """utils.py:51-81 (`batch_tensor_embeddings`).

    state      = [emb(i_0..i_{F-1}) flattened | r_0..r_{F-1}]      float32[B, F*E+F]
    next_state = [emb(i_1..i_F)     flattened | r_1..r_F]          float32[B, F*E+F]
    action     = emb(i_F)                                          float32[B, E]
    reward     = r_F                                               float32[B]
    done       = 0 except done[cumsum(sizes - F) - 1] = 1          float32[B]   (:70-71)
    """
    b = ratings.shape[0]
    emb = table[items]  # [B, F+1, E]   (utils.py:57)
    state = np.concatenate([emb[:, :-1, :].reshape(b, -1), ratings[:, :-1]], 1)
    next_state = np.concatenate([emb[:, 1:, :].reshape(b, -1), ratings[:, 1:]], 1)
    action = emb[:, -1, :].copy()
    reward = ratings[:, -1].copy()
    done = np.zeros(b, dtype=np.float32)
    if b:
        done[np.cumsum(sizes - frame_size) - 1] = 1.0
    return {
        "state": np.ascontiguousarray(state, dtype=np.float32),
        "action": np.ascontiguousarray(action, dtype=np.float32),
        "reward": reward.astype(np.float32),
        "next_state": np.ascontiguousarray(next_state, dtype=np.float32),
        "done": done,
    }


def frame_batch(user_items, user_ratings, table, frame_size: int, rows: Optional[int] = None):
    """Whole collate: windows -> gather -> SARS' dict (utils.py:161-187 + :51-81).

    `rows` is the fixed-row extension used by the benchmark (`FrameEnv(rows_per_batch=)`):
    the batch is built from the given users exactly as the reference would, then cut to
    its first `rows` rows (`done` is computed before the cut).
    """
    items, ratings, sizes = frame_windows(user_items, user_ratings, frame_size)
    out = embed_frames(items, ratings, sizes, np.asarray(table), frame_size)
    out["sizes"] = sizes
    out["items"] = items
    if rows is not None:
        for k in ("state", "action", "reward", "next_state", "done", "items"):
            out[k] = out[k][:rows]
    return out


# --------------------------------------------------------------------------------------
# 2. Networks (torch CPU fp32; explicit dropout masks)
# --------------------------------------------------------------------------------------

NetParams = Dict[str, torch.Tensor]  # keys: w1 b1 w2 b2 w3 b3, torch [out,in] layout
PARAM_ORDER = ("w1", "b1", "w2", "b2", "w3", "b3")  # == nn.Module.parameters() order


def params_from_module(mod) -> NetParams:
    """Snapshot an Actor/Critic module (reference or ours) into an oracle param dict."""
    return {
        "w1": mod.linear1.weight.detach().cpu().float().clone(),
        "b1": mod.linear1.bias.detach().cpu().float().clone(),
        "w2": mod.linear2.weight.detach().cpu().float().clone(),
        "b2": mod.linear2.bias.detach().cpu().float().clone(),
        "w3": mod.linear3.weight.detach().cpu().float().clone(),
        "b3": mod.linear3.bias.detach().cpu().float().clone(),
    }


def clone_params(p: NetParams) -> NetParams:
    return {k: v.clone() for k, v in p.items()}


def _drop(h: torch.Tensor, mask: Optional[torch.Tensor]) -> torch.Tensor:
    """nn.Dropout(p=0.5) in train mode with an explicit keep-mask (models.py:50,69,71).

    torch computes x * mask / (1-p); with p = 0.5 that is x * mask * 2 exactly.
    mask None == eval mode (target nets, algo.py:76-77).
    """
    if mask is None:
        return h
    return h * (mask.to(h.dtype) * 2.0)


def mlp_forward(p: NetParams, x: torch.Tensor, m1=None, m2=None):
    """3-layer MLP shared by Actor (models.py:66-73) and Critic (models.py:207-213).

    Returns (out, cache) where cache = (x, h1, h2): h1/h2 are the post-dropout
    activations.  That is all backward needs: with p=0.5, h>0 <=> (pre-activation>0 and
    kept), so d(pre-activation) = d(h) * 2 * [h>0].
    """
    h1 = _drop(torch.relu(torch.addmm(p["b1"], x, p["w1"].t())), m1)
    h2 = _drop(torch.relu(torch.addmm(p["b2"], h1, p["w2"].t())), m2)
    out = torch.addmm(p["b3"], h2, p["w3"].t())
    return out, (x, h1, h2)


def actor_forward(p: NetParams, state, m1=None, m2=None):
    return mlp_forward(p, state, m1, m2)


def critic_forward(p: NetParams, state, action, m1=None, m2=None):
    """`torch.cat([state, action], 1)` then the MLP (models.py:207-213)."""
    return mlp_forward(p, torch.cat([state, action], 1), m1, m2)


def mlp_backward(p: NetParams, cache, dout: torch.Tensor, train: bool = True, need_dx: bool = False,
                 need_dw: bool = True):
    """Hand-written backward of `mlp_forward`.

    dout: gradient w.r.t. the layer-3 output [B, N3].
    Returns (grads dict or None, dx or None, intermediates dict).
    """
    x, h1, h2 = cache
    s = 2.0 if train else 1.0
    g: Dict[str, torch.Tensor] = {}
    if need_dw:
        g["w3"] = dout.t() @ h2
        g["b3"] = dout.sum(0)
    dz2 = (dout @ p["w3"]) * ((h2 > 0).to(dout.dtype) * s)
    if need_dw:
        g["w2"] = dz2.t() @ h1
        g["b2"] = dz2.sum(0)
    dz1 = (dz2 @ p["w2"]) * ((h1 > 0).to(dout.dtype) * s)
    if need_dw:
        g["w1"] = dz1.t() @ x
        g["b1"] = dz1.sum(0)
    dx = dz1 @ p["w1"] if need_dx else None
    return (g if need_dw else None), dx, {"dz2": dz2, "dz1": dz1}


# --------------------------------------------------------------------------------------
# 3. Optimizer arithmetic, soft update, the clip quirk
# --------------------------------------------------------------------------------------


@dataclass
class AdamState:
    """State of one `torch.optim.Adam` instance over the 6 tensors of a net."""
    lr: float = 1e-3
    beta1: float = 0.9
    beta2: float = 0.999
    eps: float = 1e-8
    weight_decay: float = 0.0
    t: int = 0
    m: Dict[str, torch.Tensor] = field(default_factory=dict)
    v: Dict[str, torch.Tensor] = field(default_factory=dict)


def adam_step(p: NetParams, g: Dict[str, torch.Tensor], st: AdamState, grad_scale: float = 1.0) -> None:
    """torch.optim.Adam (non-amsgrad, L2 weight decay), in place.

    Restated from torch 2.x `_single_tensor_adam`:
        g   = g + wd * p
        m   = m + (1-b1) * (g - m)                (lerp_)
        v   = b2 * v + (1-b2) * g * g
        p  -= (lr / (1-b1^t)) * m / (sqrt(v) / sqrt(1-b2^t) + eps)
    The reference calls whichever optimizer the user put in `optimizer[...]`
    (misc.py:44, ddpg.py:93, td3.py:97,101,134); Adam is the documented substitution.
    """
    st.t += 1
    bc1 = 1.0 - st.beta1 ** st.t
    bc2 = 1.0 - st.beta2 ** st.t
    step_size = st.lr / bc1
    bc2_sqrt = math.sqrt(bc2)
    for k in PARAM_ORDER:
        grad = g[k] * grad_scale if grad_scale != 1.0 else g[k]
        if st.weight_decay != 0.0:
            grad = grad + st.weight_decay * p[k]
        if k not in st.m:
            st.m[k] = torch.zeros_like(p[k])
            st.v[k] = torch.zeros_like(p[k])
        st.m[k] += (1.0 - st.beta1) * (grad - st.m[k])
        st.v[k].mul_(st.beta2).addcmul_(grad, grad, value=1.0 - st.beta2)
        denom = st.v[k].sqrt() / bc2_sqrt + st.eps
        p[k] -= step_size * (st.m[k] / denom)


def soft_update(net: NetParams, target: NetParams, tau: float) -> None:
    """utils/misc.py:1-5: target = target*(1-tau) + net*tau, in that operand order."""
    for k in PARAM_ORDER:
        target[k] = target[k] * (1.0 - tau) + net[k] * tau


def clip_grad_quirk_scale(g: Dict[str, torch.Tensor]) -> float:
    """`clip_grad_norm_(params, max_norm=-1, norm_type=1)` (ddpg.py:92, td3.py:133).

    total = sum_i ||g_i||_1 ; coef = max_norm / (total + 1e-6) = -1/(total+1e-6);
    torch clamps coef to <= 1.0 (a negative coef passes) and multiplies every grad:
    the actor gradient is L1-normalised AND sign-flipped.  Returns the coefficient.
    """
    total = sum(float(v.abs().sum()) for v in g.values())
    return min(-1.0 / (total + 1e-6), 1.0)


# --------------------------------------------------------------------------------------
# 4. The DDPG / TD3 step
# --------------------------------------------------------------------------------------


@dataclass
class DDPGState:
    policy: NetParams
    value: NetParams
    target_policy: NetParams
    target_value: NetParams
    policy_opt: AdamState
    value_opt: AdamState
    params: Dict[str, float] = field(default_factory=lambda: {
        "gamma": 0.99, "min_value": -10, "max_value": 10, "policy_step": 10, "soft_tau": 0.001})  # algo.py:103-109

    @staticmethod
    def create(policy: NetParams, value: NetParams, policy_opt: AdamState, value_opt: AdamState) -> "DDPGState":
        # algo.py:73-81: targets are deep copies hard-synced with tau = 1.0
        return DDPGState(policy, value, clone_params(policy), clone_params(value), policy_opt, value_opt)


def _as_t(x):
    return x if isinstance(x, torch.Tensor) else torch.from_numpy(np.asarray(x))


def temporal_difference(reward, done, gamma, target):
    """misc.py:6-7."""
    return reward + (1.0 - done) * gamma * target


def ddpg_step(st: DDPGState, batch, masks: Sequence[Optional[torch.Tensor]], step: int, learn: bool = True,
              trace: Optional[dict] = None):
    """One `ddpg_update` call (ddpg.py:58-104 with value_update misc.py:25-44 inlined).

    masks: 6 keep-masks [B,H] in the reference's consumption order
           (critic L1, L2 | actor L1, L2 | critic L1, L2);  None entries = no dropout.
    Returns {"value": float, "policy": float, "step": step}.
    """
    s = _as_t(batch["state"]).float()
    a = _as_t(batch["action"]).float()
    r = _as_t(batch["reward"]).float().reshape(-1, 1)      # utils.py:269 unsqueeze(1)
    s2 = _as_t(batch["next_state"]).float()
    d = _as_t(batch["done"]).float().reshape(-1, 1)
    B = s.shape[0]
    P = st.params
    m = list(masks) + [None] * (6 - len(masks))

    # ---- value_update (misc.py:27-44)
    next_action, _ = actor_forward(st.target_policy, s2)                       # eval: no dropout
    target_value, _ = critic_forward(st.target_value, s2, next_action)
    expected = temporal_difference(r, d, P["gamma"], target_value)
    expected = torch.clamp(expected, P["min_value"], P["max_value"])          # misc.py:33-35
    value, vcache = critic_forward(st.value, s, a, m[0], m[1])
    diff = value - expected
    value_loss = (diff * diff).mean()                                          # misc.py:39
    if trace is not None:
        trace.update(next_action=next_action, target_value=target_value, expected=expected, value=value)
    if learn:
        dq = diff * (2.0 / B)
        gv, _, inter = mlp_backward(st.value, vcache, dq, train=m[0] is not None)
        if trace is not None:
            trace.update(value_grads=gv, value_dz2=inter["dz2"], value_dz1=inter["dz1"])
        adam_step(st.value, gv, st.value_opt)                                  # misc.py:42-44

    # ---- policy loss, through the ALREADY UPDATED critic (ddpg.py:78-79,87)
    gen_action, pcache = actor_forward(st.policy, s, m[2], m[3])
    q_pi, qcache = critic_forward(st.value, s, gen_action, m[4], m[5])
    policy_loss = -(q_pi.mean())
    if trace is not None:
        trace.update(gen_action=gen_action, q_pi=q_pi)

    if learn and step % P["policy_step"] == 0:                                 # ddpg.py:89
        dq_pi = torch.full_like(q_pi, -1.0 / B)
        _, dxa, _ = mlp_backward(st.value, qcache, dq_pi, train=m[4] is not None, need_dx=True, need_dw=False)
        dact = dxa[:, s.shape[1]:]                                             # gradient reaching gen_action
        gp, _, _ = mlp_backward(st.policy, pcache, dact, train=m[2] is not None)
        coef = clip_grad_quirk_scale(gp)                                       # ddpg.py:92
        if trace is not None:
            trace.update(policy_grads=gp, clip_coef=coef, dact=dact)
        adam_step(st.policy, gp, st.policy_opt, grad_scale=coef)               # ddpg.py:93
        soft_update(st.value, st.target_value, P["soft_tau"])                  # ddpg.py:95-100
        soft_update(st.policy, st.target_policy, P["soft_tau"])

    return {"value": float(value_loss), "policy": float(policy_loss), "step": step}


@dataclass
class TD3State:
    policy: NetParams
    value1: NetParams
    value2: NetParams
    target_policy: NetParams
    target_value1: NetParams
    target_value2: NetParams
    policy_opt: AdamState
    value_opt1: AdamState
    value_opt2: AdamState
    params: Dict[str, float] = field(default_factory=lambda: {
        "gamma": 0.99, "noise_std": 0.5, "noise_clip": 3, "soft_tau": 0.001, "policy_update": 10})  # algo.py:164-174

    @staticmethod
    def create(policy, value1, value2, policy_opt, value_opt1, value_opt2) -> "TD3State":
        return TD3State(policy, value1, value2, clone_params(policy), clone_params(value1), clone_params(value2),
                        policy_opt, value_opt1, value_opt2)


def td3_step(st: TD3State, batch, noise: torch.Tensor, masks: Sequence[Optional[torch.Tensor]], step: int,
             learn: bool = True, trace: Optional[dict] = None):
    """One `td3_update` call (td3.py:66-150).

    noise: the UNCLIPPED Gaussian draw `torch.normal(zeros(B,A), noise_std)` (td3.py:74);
           clipping to +-noise_clip happens here (td3.py:77).
    masks: 8 keep-masks in consumption order
           (critic1 L1,L2 | critic2 L1,L2 | actor L1,L2 | critic1 L1,L2).
    Quirks kept: no clamp of the TD target; the target policy net is never soft-updated
    (td3.py:136-141); value losses are `MSELoss` means.
    """
    s = _as_t(batch["state"]).float()
    a = _as_t(batch["action"]).float()
    r = _as_t(batch["reward"]).float().reshape(-1, 1)
    s2 = _as_t(batch["next_state"]).float()
    d = _as_t(batch["done"]).float().reshape(-1, 1)
    B = s.shape[0]
    P = st.params
    m = list(masks) + [None] * (8 - len(masks))

    next_action, _ = actor_forward(st.target_policy, s2)
    nz = torch.clamp(_as_t(noise).float(), -P["noise_clip"], P["noise_clip"])   # td3.py:77
    next_action = next_action + nz                                              # td3.py:78
    tq1, _ = critic_forward(st.target_value1, s2, next_action)
    tq2, _ = critic_forward(st.target_value2, s2, next_action)
    tq = torch.min(tq1, tq2)                                                    # td3.py:83
    expected = temporal_difference(r, d, P["gamma"], tq)                        # td3.py:84-86 (no clamp)

    q1, c1 = critic_forward(st.value1, s, a, m[0], m[1])
    q2, c2 = critic_forward(st.value2, s, a, m[2], m[3])
    d1 = q1 - expected
    d2 = q2 - expected
    loss1 = (d1 * d1).mean()
    loss2 = (d2 * d2).mean()
    if trace is not None:
        trace.update(next_action=next_action, expected=expected, q1=q1, q2=q2)
    if learn:
        g1, _, _ = mlp_backward(st.value1, c1, d1 * (2.0 / B), train=m[0] is not None)
        adam_step(st.value1, g1, st.value_opt1)                                  # td3.py:95-97
        g2, _, _ = mlp_backward(st.value2, c2, d2 * (2.0 / B), train=m[2] is not None)
        if trace is not None:
            trace.update(value1_grads=g1, value2_grads=g2)
        adam_step(st.value2, g2, st.value_opt2)                                  # td3.py:99-101

    gen_action, pcache = actor_forward(st.policy, s, m[4], m[5])
    q_pi, qcache = critic_forward(st.value1, s, gen_action, m[6], m[7])
    policy_loss = -(q_pi.mean())
    if trace is not None:
        trace.update(gen_action=gen_action, q_pi=q_pi)

    if step % P["policy_update"] == 0 and learn:                                 # td3.py:130
        dq_pi = torch.full_like(q_pi, -1.0 / B)
        _, dxa, _ = mlp_backward(st.value1, qcache, dq_pi, train=m[6] is not None, need_dx=True, need_dw=False)
        dact = dxa[:, s.shape[1]:]
        gp, _, _ = mlp_backward(st.policy, pcache, dact, train=m[4] is not None)
        coef = clip_grad_quirk_scale(gp)
        if trace is not None:
            trace.update(policy_grads=gp, clip_coef=coef, dact=dact)
        adam_step(st.policy, gp, st.policy_opt, grad_scale=coef)
        soft_update(st.value1, st.target_value1, P["soft_tau"])
        soft_update(st.value2, st.target_value2, P["soft_tau"])
        # target_policy is NOT updated (td3.py:136-141)

    return {"value1": float(loss1), "value2": float(loss2), "policy": float(policy_loss), "step": step}


# --------------------------------------------------------------------------------------
# 5. RNG recipes that reproduce the reference's CPU draws (verified in make_golden.py)
# --------------------------------------------------------------------------------------


def draw_dropout_masks(n: int, batch: int, hidden: int) -> List[torch.Tensor]:
    """`nn.Dropout(0.5)` in train mode on CPU consumes the default generator as
    `torch.empty_like(x).bernoulli_(0.5)` once per call, in call order."""
    return [torch.empty(batch, hidden).bernoulli_(0.5).to(torch.uint8) for _ in range(n)]


def draw_td3_noise(batch: int, action_dim: int, std: float) -> torch.Tensor:
    """`torch.normal(torch.zeros(B,A), std)` (td3.py:74) == `torch.randn(B,A) * std` bit-for-bit."""
    return torch.normal(torch.zeros(batch, action_dim), std)
