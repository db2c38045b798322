"""Shared helpers for the test-suite (synthetic replay stores, CSR builders, tolerances)."""
import numpy as np
import torch


def make_store(n_users, n_items, emb_dim, min_len, max_len, seed):
    rng = np.random.default_rng(seed)
    lens = rng.integers(min_len, max_len + 1, size=n_users)
    items = [rng.integers(0, n_items, size=int(L)).astype(np.int64) for L in lens]
    ratings = [(2.0 * (rng.integers(1, 11, size=int(L)) * 0.5 - 2.5)).astype(np.float64) for L in lens]
    table = rng.standard_normal((n_items, emb_dim)).astype(np.float32)
    return items, ratings, table


def csr(items, ratings):
    off = np.zeros(len(items) + 1, dtype=np.int64)
    off[1:] = np.cumsum([len(i) for i in items])
    return (np.concatenate(items).astype(np.int32), np.concatenate(ratings).astype(np.float32), off)


def rel_err(a, b):
    a = a.detach().double().cpu() if isinstance(a, torch.Tensor) else torch.as_tensor(np.asarray(a), dtype=torch.float64)
    b = b.detach().double().cpu() if isinstance(b, torch.Tensor) else torch.as_tensor(np.asarray(b), dtype=torch.float64)
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def fro_err(a, b):
    """Relative Frobenius-norm error ||a-b|| / ||b|| (robust to a handful of outlying elements)."""
    a = a.detach().double().cpu() if isinstance(a, torch.Tensor) else torch.as_tensor(np.asarray(a), dtype=torch.float64)
    b = b.detach().double().cpu() if isinstance(b, torch.Tensor) else torch.as_tensor(np.asarray(b), dtype=torch.float64)
    return float((a - b).norm() / (b.norm() + 1e-30))


# ---- measured-vs-asserted bookkeeping for the bf16 parity bounds ("bounds 30x looser than measured").  Every bf16
# tolerance check goes through `within`: it asserts value <= bound AND records the largest value ever seen under that name in
# gpurun_out/measured_bounds.json.  The bound of a bf16 quantity comes from BF16_BOUNDS below: 2.5x the value measured on an
# MI355X (profiles/r03_measured_bounds.json is the recorder's file from that run), never above the round-2 constant;
# tests/test_bounds_table.py checks table against file (every bound <= 3x its measured value).
import json as _json
import os as _os

_BOUNDS_FILE = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "gpurun_out", "measured_bounds.json")

BF16_BOUNDS = {
    "bench_shape/bf16/loss_curve_policy": 4.1e-03,   # measured 1.63e-03
    "bench_shape/bf16/loss_curve_value": 5.7e-04,   # measured 2.27e-04
    "ddpg_vs_oracle/bf16/clip_coef": 2.9e-04,   # measured 1.14e-04
    "ddpg_vs_oracle/bf16/dact": 9.2e-03,   # measured 3.66e-03
    "ddpg_vs_oracle/bf16/dz1": 6.8e-03,   # measured 2.71e-03
    "ddpg_vs_oracle/bf16/dz2": 3.9e-03,   # measured 1.54e-03
    "ddpg_vs_oracle/bf16/fwd/expected": 3.1e-03,   # measured 1.22e-03
    "ddpg_vs_oracle/bf16/fwd/gen_action": 1.1e-02,   # measured 4.03e-03
    "ddpg_vs_oracle/bf16/fwd/next_action": 1.2e-02,   # measured 4.47e-03
    "ddpg_vs_oracle/bf16/fwd/q1": 1.0e-02,   # measured 3.98e-03
    "ddpg_vs_oracle/bf16/fwd/target_q": 8.9e-03,   # measured 3.55e-03
    "ddpg_vs_oracle/bf16/loss": 2.4e-02,   # measured 9.30e-03
    "ddpg_vs_oracle/bf16/params/policy": 1.8e-02,   # measured 6.85e-03
    "ddpg_vs_oracle/bf16/params/target_policy": 1.7e-05,   # measured 6.49e-06
    "ddpg_vs_oracle/bf16/params/target_value": 4.5e-05,   # measured 1.77e-05
    "ddpg_vs_oracle/bf16/params/value": 5.0e-02,   # measured 2.41e-02
    "ddpg_vs_oracle/bf16/policy_grad": 1.3e-02,   # measured 5.15e-03
    "ddpg_vs_oracle/bf16/value_grad": 9.0e-03,   # measured 3.57e-03
    "reinforce_beta_100k/bf16/loss": 3.9e-03,   # measured 1.55e-03 (round 5)
    "reinforce_beta_100k/bf16/params/beta": 1.2e-03,   # measured 4.92e-04 (round 5)
    "reinforce_beta_100k/bf16/params/policy": 9.2e-02,   # measured 3.68e-02 (round 5)
    "reinforce_beta_100k/bf16/params/target_policy": 1.5e-04,   # measured 6.02e-05 (round 5)
    "reinforce_beta_100k/bf16/params/target_value": 1.5e-05,   # measured 5.85e-06 (round 5)
    "reinforce_beta_100k/bf16/params/value": 1.2e-03,   # measured 4.95e-04 (round 5)
    "td3_vs_oracle/bf16/loss": 6.7e-03,   # measured 2.64e-03
    "td3_vs_oracle/bf16/params/policy": 5.0e-02,   # measured 2.62e-02
    "td3_vs_oracle/bf16/params/target_policy": 1.0e-06,   # measured 0.00e+00
    "td3_vs_oracle/bf16/params/target_value1": 4.5e-05,   # measured 1.79e-05
    "td3_vs_oracle/bf16/params/value1": 5.0e-02,   # measured 2.15e-02
    "td3_vs_oracle/bf16/params/value2": 5.0e-02,   # measured 2.15e-02
}


def within(name, value, bound=None):
    value = float(value)
    bound = float(BF16_BOUNDS.get(name, bound))
    try:
        _os.makedirs(_os.path.dirname(_BOUNDS_FILE), exist_ok=True)
        rec = _json.load(open(_BOUNDS_FILE)) if _os.path.exists(_BOUNDS_FILE) else {}
        old = rec.get(name, {"measured": 0.0})
        rec[name] = {"measured": max(float(old["measured"]), value), "bound": bound}
        _json.dump(rec, open(_BOUNDS_FILE, "w"), indent=1, sort_keys=True)
    except OSError:
        pass
    assert value <= bound, (name, value, bound)


# ---- split-bf16 rows (RECNN_BF16X3, recnn_amd/csrc/x3.h) restated in torch for the tests: logical column c of a row lives at
# physical column 2 (c & ~31) + (c & 31) (hi = bf16(x)) and 32 elements further (lo = bf16(x - hi))
def x3_cols(n):
    c = torch.arange(n)
    return (c // 32) * 64 + (c % 32)


def x3_pack_ref(x, ld=None):
    """fp32 [R, C] -> bfloat16 [R, ld] in the split layout (padding columns zero)."""
    x = x.float()
    R, Cc = x.shape
    ldp = 2 * ((Cc + 31) // 32 * 32) if ld is None else ld
    hi = x.bfloat16()
    lo = (x - hi.float()).bfloat16()
    out = torch.zeros(R, ldp, dtype=torch.bfloat16)
    col = x3_cols(Cc)
    out[:, col] = hi
    out[:, col + 32] = lo
    return out


def x3_unpack_ref(xp, cols):
    col = x3_cols(cols).to(xp.device)
    return xp[:, col].float() + xp[:, col + 32].float()


def x3_value(x):
    """what a split row holds for x: hi + lo (fp32)"""
    hi = x.float().bfloat16().float()
    return hi + (x.float() - hi).bfloat16().float()


class TorchOps:
    """Stand-in for recnn_amd.parallel._HipOps in the gloo tests on CPU (the vocab-parallel modules take `ops=`): the three
    products in torch.  Test infrastructure -- the product package has no CPU implementation of its ops."""
    import torch as _t
    linear = staticmethod(lambda x, w, b, relu: TorchOps._t.relu(x @ w.t() + b) if relu else x @ w.t() + b)
    grad_w = staticmethod(lambda dz, x: dz.t() @ x)
    grad_x = staticmethod(lambda dz, w: dz @ w)
    rowmax = staticmethod(lambda x: x.amax(1))

    @staticmethod
    def exp_rowsum_(x, m):
        x.sub_(m[:, None]).exp_()
        return x.sum(1)

    @staticmethod
    def norm_pick_(x, ssum, local):
        x.div_(ssum[:, None])
        ns = x.shape[1]
        own = (local >= 0) & (local < ns)
        t = TorchOps._t
        return t.where(own, x.gather(1, local.clamp(0, ns - 1)[:, None])[:, 0], t.zeros_like(ssum))

    @staticmethod
    def logprob_bwd(probs, local, g):
        dlog = probs * (-g)[:, None]
        ns = probs.shape[1]
        rows = ((local >= 0) & (local < ns)).nonzero()[:, 0]
        dlog[rows, local[rows]] += g[rows]
        return dlog
