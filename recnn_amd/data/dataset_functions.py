"""One-time ratings-CSV ETL producing the replay-store contract (reference: recnn/data/dataset_functions.py).

Out of the hot path: only its OUTPUT matters to the step --
  user_dict[user_id] = {"items": int64[L] dense item ids in time order, "ratings": float64[L] = 2*(r-2.5)}
  users = ids of the users with more than `frame_size` interactions, longest history first.
"""
from typing import Callable, Dict, List

import numpy as np

from .pandas_backend import pd

__all__ = ["DataFuncKwargs", "DataFuncArgsMut", "prepare_dataset", "prepare_dataset_device", "truncate_dataset",
           "build_data_pipeline", "csr_from_ratings", "csr_from_ratings_device"]


class DataFuncKwargs:
    """Immutable-by-convention keyword bag handed down a data pipeline (dataset_functions.py:48-70)."""

    def __init__(self, **kwargs):
        self.kwargs = kwargs

    def keys(self):
        return self.kwargs.keys()

    def get(self, name: str):
        if name not in self.kwargs:
            raise AttributeError(
                f"No kwarg with name {name} found! Set it at the top of your prepare_dataset function: kwargs.set('{name}', value)")
        return self.kwargs[name]

    def set(self, name: str, value):
        self.kwargs[name] = value


class DataFuncArgsMut:
    """Mutable pipeline state (dataset_functions.py:73-81)."""

    def __init__(self, df, base, users: List[int], user_dict: Dict[int, Dict[str, np.ndarray]]):
        self.base = base
        self.users = users
        self.user_dict = user_dict
        self.df = df
        self.csr = None


def csr_from_ratings(user_ids: np.ndarray, item_ids: np.ndarray, ratings: np.ndarray, timestamps: np.ndarray):
    """One pass from ratings rows to the replay store's CSR: rows ordered by (user, time), as the reference's
    `df.sort_values(by="timestamp")` + `groupby("userId")` leaves them (dataset_functions.py:103-121).

    pandas' `sort_values` is numpy's default (unstable) argsort of the timestamp column and `groupby` keeps the row
    order inside a group, so the reference's order -- including the order of rows with EQUAL timestamps -- is
    `argsort(ts)` followed by a stable partition by user; reproduced here with two argsorts and no Python loop.
    Returns (users sorted ascending, user_off int64[U+1], items int64[n], ratings float64[n])."""
    by_time = np.argsort(timestamps)                         # kind="quicksort": what DataFrame.sort_values uses
    order = by_time[_stable_argsort_ids(user_ids[by_time])]
    u_sorted = user_ids[order]
    first = np.flatnonzero(np.concatenate(([True], u_sorted[1:] != u_sorted[:-1]))) if len(order) else np.zeros(0, np.int64)
    users = u_sorted[first]
    user_off = np.empty(len(users) + 1, dtype=np.int64)
    user_off[:-1] = first
    user_off[-1] = len(order)
    return users, user_off, item_ids[order], ratings[order]


def csr_from_ratings_device(user_ids, item_keys, ratings, timestamps, key_to_id=None, device="cuda", want_mapped_rows=False):
    """`csr_from_ratings` on the GPU (csrc/csr.hip, `recnn_csr_build`): composite (user, timestamp) key, stable LSD radix
    sort, gather, first-row flags + scan -- with the two per-row transforms of `prepare_dataset` folded into the gather
    (rating 2 (r - 2.5) in fp64; item key -> dense id through `key_to_id` by binary search).  Takes the RAW columns.

    Order of rows with equal (user, timestamp): input order (pandas' unstable sort leaves it host-dependent in the
    reference); everything else is bit-identical to `csr_from_ratings`.  No CPU fallback: raises without the HIP library
    or a GPU.  Returns (users, user_off, items int64, ratings float64) as numpy arrays [+ the mapped id column in input
    row order when `want_mapped_rows`], or None if some item key is missing from `key_to_id` (the caller then takes the
    host path, which reproduces pandas' NaN semantics)."""
    import ctypes as C
    import torch
    from .. import _lib as L
    if not torch.cuda.is_available():
        raise L.RecnnHipError("csr_from_ratings_device: needs a GPU (use csr_from_ratings for the host path)")
    n = len(user_ids)
    if n == 0:
        e = np.zeros(0, dtype=np.int64)
        out = (e, np.zeros(1, dtype=np.int64), e.copy(), np.zeros(0, dtype=np.float64))
        return out + (e.copy(),) if want_mapped_rows else out
    dev = torch.device(device)
    up = lambda a, dt: torch.from_numpy(np.ascontiguousarray(np.asarray(a), dtype=dt)).to(dev)
    d_user, d_item, d_ts = up(user_ids, np.int64), up(item_keys, np.int64), up(timestamps, np.int64)
    d_rating = up(ratings, np.float64)
    mk = mv = None
    n_map = 0
    if key_to_id is not None:
        keys = np.fromiter(key_to_id.keys(), dtype=np.int64, count=len(key_to_id))
        vals = np.fromiter(key_to_id.values(), dtype=np.int64, count=len(key_to_id))
        o = np.argsort(keys)
        mk, mv, n_map = up(keys[o], np.int64), up(vals[o], np.int64), len(keys)
    need = C.c_int64(0)
    L.call("recnn_csr_workspace_bytes", n, C.byref(need))
    ws = torch.empty(need.value, dtype=torch.uint8, device=dev)
    items = torch.empty(n, dtype=torch.int64, device=dev)
    rates = torch.empty(n, dtype=torch.float64, device=dev)
    users = torch.empty(n, dtype=torch.int64, device=dev)
    off = torch.empty(n + 1, dtype=torch.int64, device=dev)
    mapped = torch.empty(n, dtype=torch.int64, device=dev) if want_mapped_rows else None
    counts = (C.c_int64 * 3)()
    with torch.cuda.device(dev):
        L.call("recnn_csr_build", L.ptr(d_user), L.ptr(d_item), L.ptr(d_rating), L.ptr(d_ts), n, L.ptr(mk), L.ptr(mv), n_map,
               L.ptr(items), L.ptr(rates), L.ptr(users), L.ptr(off), None, L.ptr(mapped), counts, L.ptr(ws), need.value,
               L.current_stream())
    n_users, missing = int(counts[0]), int(counts[1])
    if missing:
        return None
    out = (users[:n_users].cpu().numpy(), off[:n_users + 1].cpu().numpy(), items.cpu().numpy(), rates.cpu().numpy())
    return out + (mapped.cpu().numpy(),) if want_mapped_rows else out


def _stable_argsort_ids(x: np.ndarray) -> np.ndarray:
    """Stable argsort of non-negative integer ids below 2^32 as two 16-bit radix passes (numpy's stable sort is a radix
    sort for 16-bit keys, a merge sort otherwise: 3x slower on 20M rows)."""
    if len(x) == 0 or not np.issubdtype(x.dtype, np.integer) or x.min() < 0 or x.max() >= (1 << 32):
        return np.argsort(x, kind="stable")
    p1 = np.argsort((x & 0xFFFF).astype(np.uint16), kind="stable")
    if x.max() < (1 << 16):
        return p1
    p2 = np.argsort((x[p1] >> 16).astype(np.uint16), kind="stable")
    return p1[p2]


def _map_keys(col, key_to_id):
    """`col.map(key_to_id)` without a Python dict lookup per row: binary search over the sorted keys.  Falls back to
    `Series.map` when a row's key is missing from the dict (the reference then yields NaN there)."""
    keys = np.fromiter(key_to_id.keys(), dtype=np.int64, count=len(key_to_id))
    vals = np.fromiter(key_to_id.values(), dtype=np.int64, count=len(key_to_id))
    order = np.argsort(keys)
    keys, vals = keys[order], vals[order]
    x = col.to_numpy()
    if len(keys) == 0 or not np.issubdtype(x.dtype, np.integer):
        return col.map(key_to_id)
    pos = np.minimum(np.searchsorted(keys, x), len(keys) - 1)
    if not np.array_equal(keys[pos], x):
        return col.map(key_to_id)
    return vals[pos]


def prepare_dataset(args_mut: DataFuncArgsMut, kwargs: DataFuncKwargs):
    """ratings frame (userId, movieId, rating, timestamp) -> user_dict / users (dataset_functions.py:84-126).

    Vectorised: the reference applies a Python lambda per row for the rating transform and the id map and a Python
    callback per user for the grouping; here these are three array passes and `csr_from_ratings`.  `user_dict[uid]`
    holds views into the two sorted arrays (the CSR the device store uploads as is, `args_mut.csr`)."""
    frame_size = kwargs.get("frame_size")
    key_to_id = args_mut.base.key_to_id
    df = args_mut.df
    df["rating"] = 2.0 * (df["rating"] - 2.5)                 # [0.5, 5] -> [-4, 5]
    df["movieId"] = _map_keys(df["movieId"], key_to_id)       # sparse movie ids -> dense table rows
    if pd.get_type() == "modin":
        df = df._to_pandas()
    uid = df["userId"].to_numpy()
    users_sorted, user_off, items, ratings = csr_from_ratings(uid, df["movieId"].to_numpy(), df["rating"].to_numpy(),
                                                              df["timestamp"].to_numpy())
    lens = np.diff(user_off)
    # users with more than frame_size ratings, longest history first (ties: pandas' sort_values order of the counts)
    counts = df.groupby("userId").size()
    users = counts[counts > frame_size].sort_values(ascending=False).index
    user_dict = {u: {"items": items[a:b], "ratings": ratings[a:b]}
                 for u, a, b in zip(users_sorted.tolist(), user_off[:-1].tolist(), user_off[1:].tolist())}
    assert len(lens) == len(user_dict)
    args_mut.df = df
    args_mut.user_dict = user_dict
    args_mut.users = users
    args_mut.csr = (users_sorted, user_off, items, ratings)
    return args_mut, kwargs


def prepare_dataset_device(args_mut: DataFuncArgsMut, kwargs: DataFuncKwargs):
    """`prepare_dataset` with the sort / map / transform / grouping on the GPU (`csr_from_ratings_device`); same outputs
    (`user_dict`, `users`, `csr`, the mutated frame).  Pass it as `FrameEnv(..., prepare_dataset=prepare_dataset_device)`.
    Differs from `prepare_dataset` only in the order of rows that share (user, timestamp): input order here (see
    csrc/csr.hip).  Frames holding an item key that `key_to_id` lacks go through the host path."""
    frame_size = kwargs.get("frame_size")
    df = args_mut.df
    if pd.get_type() == "modin":
        df = df._to_pandas()
    res = csr_from_ratings_device(df["userId"].to_numpy(), df["movieId"].to_numpy(), df["rating"].to_numpy(),
                                  df["timestamp"].to_numpy(), key_to_id=args_mut.base.key_to_id, want_mapped_rows=True)
    if res is None:
        return prepare_dataset(args_mut, kwargs)
    users_sorted, user_off, items, ratings, mapped = res
    df["rating"] = 2.0 * (df["rating"] - 2.5)
    df["movieId"] = mapped
    lens = np.diff(user_off)
    # groupby("userId").size() is this Series (ascending user ids): the same sort_values call as the host path then
    # orders users with equal counts the same way
    counts = pd.get().Series(lens, index=pd.get().Index(users_sorted, name="userId"))
    users = counts[counts > frame_size].sort_values(ascending=False).index
    user_dict = {u: {"items": items[a:b], "ratings": ratings[a:b]}
                 for u, a, b in zip(users_sorted.tolist(), user_off[:-1].tolist(), user_off[1:].tolist())}
    args_mut.df = df
    args_mut.user_dict = user_dict
    args_mut.users = users
    args_mut.csr = (users_sorted, user_off, items, ratings)
    return args_mut, kwargs


def truncate_dataset(args_mut: DataFuncArgsMut, kwargs: DataFuncKwargs):
    """Keep only the `reduce_items_to` most rated items and re-densify the ids (dataset_functions.py:129-167)."""
    num_items = kwargs.get("reduce_items_to")
    df = args_mut.df
    counts = df["movieId"].value_counts().sort_values()
    keep = set(counts.index[-num_items:])
    old_key_to_id = args_mut.base.key_to_id
    keep_rows = np.zeros(len(old_key_to_id), dtype=bool)
    new_key_to_id, new_id_to_key = {}, {}
    for key, old_id in old_key_to_id.items():
        if key in keep:
            new_id = len(new_key_to_id)
            new_key_to_id[key] = new_id
            new_id_to_key[new_id] = key
            keep_rows[old_id] = True
    args_mut.df = df[df["movieId"].isin(keep)].copy()
    args_mut.base.embeddings = args_mut.base.embeddings[keep_rows]
    args_mut.base.key_to_id = new_key_to_id
    args_mut.base.id_to_key = new_id_to_key
    print(f"action space is reduced to {len(counts)} - {len(counts) - num_items} = {num_items}")
    return args_mut, kwargs


def build_data_pipeline(chain: List[Callable], kwargs: DataFuncKwargs, args_mut: DataFuncArgsMut):
    for call in chain:
        args_mut, _ = call(args_mut, kwargs)
    return args_mut, kwargs
