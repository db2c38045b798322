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
