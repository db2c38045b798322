"""Device-resident replay store: the CSR form of the reference's `user_dict`.

items   int32[sum L]   dense item ids, time-sorted per user      (reference: int64 numpy arrays per user)
ratings float32[sum L] 2*(r-2.5)                                 (reference casts float64 -> float32 per batch,
                                                                   recnn/data/utils.py:178; the cast is value-identical)
user_off int64[U+1]
Lives in HBM for the whole run (ML20M: ~160 MB); nothing is re-uploaded per batch.
"""
from typing import Dict, Sequence

import numpy as np
import torch


class ReplayStore:
    def __init__(self, user_ids: Sequence, user_dict: Dict, device: torch.device):
        self.user_ids = list(user_ids)
        self.slot_of = {u: i for i, u in enumerate(self.user_ids)}
        lens = np.fromiter((len(user_dict[u]["items"]) for u in self.user_ids), dtype=np.int64, count=len(self.user_ids))
        off = np.zeros(len(lens) + 1, dtype=np.int64)
        off[1:] = np.cumsum(lens)
        items = np.empty(int(off[-1]), dtype=np.int32)
        ratings = np.empty(int(off[-1]), dtype=np.float32)
        for i, u in enumerate(self.user_ids):
            items[off[i]:off[i + 1]] = user_dict[u]["items"]
            ratings[off[i]:off[i + 1]] = user_dict[u]["ratings"]
        self.lengths = lens                       # host copy: batch row counts are known without a device sync
        self.device = device
        self.items = torch.from_numpy(items).to(device)
        self.ratings = torch.from_numpy(ratings).to(device)
        self.user_off = torch.from_numpy(off).to(device)

    @classmethod
    def from_arrays(cls, items: np.ndarray, ratings: np.ndarray, user_off: np.ndarray, device: torch.device):
        self = cls.__new__(cls)
        n = len(user_off) - 1
        self.user_ids = list(range(n))
        self.slot_of = None
        self.lengths = np.diff(user_off).astype(np.int64)
        self.device = device
        self.items = torch.from_numpy(np.ascontiguousarray(items, dtype=np.int32)).to(device)
        self.ratings = torch.from_numpy(np.ascontiguousarray(ratings, dtype=np.float32)).to(device)
        self.user_off = torch.from_numpy(np.ascontiguousarray(user_off, dtype=np.int64)).to(device)
        return self

    def slots(self, user_ids) -> np.ndarray:
        if self.slot_of is None:
            return np.asarray(user_ids, dtype=np.int32)
        return np.fromiter((self.slot_of[u] for u in user_ids), dtype=np.int32, count=len(user_ids))

    def __len__(self):
        return len(self.lengths)
