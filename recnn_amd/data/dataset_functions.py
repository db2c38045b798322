"""One-time ratings-CSV ETL producing the replay-store contract (reference: recnn/data/dataset_functions.py).

Out of the hot path: only its OUTPUT matters to the step --
  user_dict[user_id] = {"items": int64[L] dense item ids in time order, "ratings": float64[L] = 2*(r-2.5)}
  users = ids of the users with more than `frame_size` interactions, longest history first.
"""
from typing import Callable, Dict, List

import numpy as np

from .pandas_backend import pd

__all__ = ["DataFuncKwargs", "DataFuncArgsMut", "prepare_dataset", "truncate_dataset", "build_data_pipeline"]


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


def prepare_dataset(args_mut: DataFuncArgsMut, kwargs: DataFuncKwargs):
    """ratings frame (userId, movieId, rating, timestamp) -> user_dict / users (dataset_functions.py:84-126)."""
    frame_size = kwargs.get("frame_size")
    key_to_id = args_mut.base.key_to_id
    df = args_mut.df
    df["rating"] = 2.0 * (df["rating"] - 2.5)                 # [0.5, 5] -> [-4, 5]
    df["movieId"] = df["movieId"].map(key_to_id)              # sparse movie ids -> dense table rows
    counts = df.groupby("userId").size()
    users = counts[counts > frame_size].sort_values(ascending=False).index
    if pd.get_type() == "modin":
        df = df._to_pandas()
    ordered = df.sort_values(by="timestamp")
    user_dict = {}
    for uid, grp in ordered.groupby("userId", sort=False):
        user_dict[uid] = {"items": grp["movieId"].values, "ratings": grp["rating"].values}
    args_mut.df = df
    args_mut.user_dict = user_dict
    args_mut.users = users
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
