"""FrameEnv: static-length user environment producing SARS' batches (reference: recnn/data/env.py:23-256).

Same constructor, attributes and batch dict as the reference; the data path underneath is MI355X-first:
the user histories live in HBM as a CSR replay store, a batch is described by a list of user slots, and one
HIP kernel builds state / next_state / action / reward / done directly in the packed rows the MFMA GEMMs read.
No DataLoader worker processes, no pickling of batches, no host-to-device copy of 22 MB per step.
"""
import os
import pickle

import numpy as np
import torch

from . import dataset_functions as dset_F
from . import utils
from .pandas_backend import pd
from .store import ReplayStore

__all__ = ["UserDataset", "EnvBase", "DataPath", "Env", "FrameEnv", "FrameLoader"]


class UserDataset:
    """user index -> {'items', 'rates', 'sizes', 'users'} (env.py:23-64).  Pickle-compatible field names."""

    def __init__(self, users, user_dict):
        self.users = users
        self.user_dict = user_dict

    def __len__(self):
        return len(self.users)

    def __getitem__(self, idx):
        uid = self.users[idx]
        group = self.user_dict[uid]
        items = group["items"][:]
        return {"items": items, "rates": group["ratings"][:], "sizes": items.shape[0], "users": uid}


class EnvBase:
    """What gets pickled as the env cache (env.py:67-78)."""

    def __init__(self):
        self.train_user_dataset = None
        self.test_user_dataset = None
        self.embeddings = None
        self.key_to_id = None
        self.id_to_key = None


class DataPath:
    """Paths of ratings csv / embeddings pickle / optional cache (env.py:81-98)."""

    def __init__(self, base: str, ratings: str, embeddings: str, cache: str = "", use_cache: bool = True):
        self.ratings = base + ratings
        self.embeddings = base + embeddings
        self.cache = base + cache
        self.use_cache = use_cache


class Env:
    """Builds or loads the EnvBase (env.py:101-187)."""

    def __init__(self, path: DataPath, prepare_dataset=dset_F.prepare_dataset, embed_batch=utils.batch_tensor_embeddings,
                 **kwargs):
        self.base = EnvBase()
        self.embed_batch = embed_batch
        self.prepare_dataset = prepare_dataset
        if path is None:
            return
        if path.use_cache and os.path.isfile(path.cache):
            self.load_env(path.cache)
        else:
            self.process_env(path)
            if path.use_cache:
                self.save_env(path.cache)

    def process_env(self, path: DataPath, **kwargs):
        # NB (reference quirk, env.py:137-150): called without kwargs, so the user filter always uses
        # frame_size=10 / test_size=0.05 whatever the constructor was given.
        frame_size = kwargs.get("frame_size", 10)
        test_size = kwargs.get("test_size", 0.05)
        with open(path.embeddings, "rb") as f:
            key_dict = pickle.load(f)
        self.base.embeddings, self.base.key_to_id, self.base.id_to_key = utils.make_items_tensor(key_dict)
        ratings = pd.get().read_csv(path.ratings)
        args_mut = dset_F.DataFuncArgsMut(df=ratings, base=self.base, users=None, user_dict=None)
        self.prepare_dataset(args_mut, dset_F.DataFuncKwargs(frame_size=frame_size))
        self.base = args_mut.base
        self.df = args_mut.df
        self._split(args_mut.users, args_mut.user_dict, test_size)

    def _split(self, users, user_dict, test_size):
        from sklearn.model_selection import train_test_split
        train_users, test_users = train_test_split(users, test_size=test_size)
        train_users = utils.sort_users_itemwise(user_dict, train_users)[2:]   # env.py:178 drops the 2 longest
        test_users = utils.sort_users_itemwise(user_dict, test_users)
        self.base.train_user_dataset = UserDataset(train_users, user_dict)
        self.base.test_user_dataset = UserDataset(test_users, user_dict)

    def load_env(self, where: str):
        with open(where, "rb") as f:
            self.base = pickle.load(f)

    def save_env(self, where: str):
        with open(where, "wb") as f:
            pickle.dump(self.base, f)


class FrameLoader:
    """What `env.train_dataloader` is: a re-iterable, len()-able, shuffled stream of batches of
    `batch_size` USERS (reference: torch DataLoader(shuffle=True, collate_fn=...), env.py:225-239).
    A new random user order is drawn per iteration from torch's global CPU generator, as RandomSampler does."""

    def __init__(self, env, dataset: UserDataset, batch_size: int, shuffle: bool = True):
        self.env = env
        self.dataset = dataset
        self.batch_size = batch_size
        self.shuffle = shuffle

    def __len__(self):
        return (len(self.dataset) + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        planner = getattr(self, "planner", None)
        if planner is not None:
            # an Algo drives this loader (Algo.attach_env(..., drive_loader=True)): one epoch of handles of the fixed-size
            # batches its engine draws -- `algo.update(batch)` queues them, `batch["state"]` materialises one on demand
            yield from planner.batches(planner.batches_left_in_epoch())
            return
        n = len(self.dataset)
        order = torch.randperm(n).numpy() if self.shuffle else np.arange(n)
        users = self.dataset.users
        for i in range(0, n, self.batch_size):
            yield self.env.collate_users([users[j] for j in order[i:i + self.batch_size]])


class FrameEnv(Env):
    """Static length user environment (env.py:190-256).

    Extra keyword arguments (extensions, all optional):
      device          torch device of the replay store and the batches (default: cuda)
      rows_per_batch  cut every batch to exactly this many transition rows (benchmark configuration
                      "batch 2048"); the reference's batch_size counts USERS, so its row count varies
      contiguous      return reference-layout contiguous tensors instead of views into packed rows
    """

    def __init__(self, path, frame_size=10, batch_size=25, num_workers=1, *args, device=None, rows_per_batch=None,
                 contiguous=False, **kwargs):
        kwargs["frame_size"] = frame_size
        super().__init__(path, min_seq_size=frame_size + 1, *args, **kwargs)
        self.frame_size = frame_size
        self.batch_size = batch_size
        self.num_workers = num_workers          # kept for API compatibility; there are no worker processes
        self.rows_per_batch = rows_per_batch
        self.contiguous = contiguous
        self.device = torch.device("cuda" if device is None else device)
        self._store = None
        self._table = None
        if path is not None:
            self._make_loaders()

    @classmethod
    def from_user_dict(cls, embeddings: torch.Tensor, user_dict, train_users, test_users=(), frame_size=10, batch_size=25,
                       **kwargs):
        """Build an env from in-memory data (the output contract of `prepare_dataset`) without csv / pickle files."""
        self = cls(None, frame_size, batch_size, **kwargs)
        self.base.embeddings = embeddings
        n = embeddings.shape[0]
        self.base.key_to_id = {i: i for i in range(n)}
        self.base.id_to_key = {i: i for i in range(n)}
        self.base.train_user_dataset = UserDataset(list(train_users), user_dict)
        self.base.test_user_dataset = UserDataset(list(test_users), user_dict)
        self._make_loaders()
        return self

    @classmethod
    def from_store(cls, embeddings: torch.Tensor, items, ratings, user_off, frame_size=10, batch_size=25, test_fraction=0.05,
                   **kwargs):
        """Build an env straight from CSR arrays (items int[sum L], ratings float[sum L], user_off int64[U+1]): the
        replay-store form of the reference's `user_dict`, for data that never existed as per-user python objects."""
        self = cls(None, frame_size, batch_size, **kwargs)
        self.base.embeddings = embeddings
        n_users = len(user_off) - 1
        n_test = int(n_users * test_fraction)
        ids = list(range(n_users))
        self._csr = (np.asarray(items), np.asarray(ratings), np.asarray(user_off, dtype=np.int64))
        self.base.train_user_dataset = UserDataset(ids[: n_users - n_test], None)
        self.base.test_user_dataset = UserDataset(ids[n_users - n_test:], None)
        self._make_loaders()
        return self

    def _make_loaders(self):
        self.train_dataloader = FrameLoader(self, self.base.train_user_dataset, self.batch_size, shuffle=True)
        self.test_dataloader = FrameLoader(self, self.base.test_user_dataset, self.batch_size, shuffle=True)

    # ------------------------------------------------------------------ device store
    @property
    def store(self) -> ReplayStore:
        if self._store is None:
            if self.device.type != "cuda" or not torch.cuda.is_available():
                from .. import _lib as L
                raise L.RecnnHipError(f"FrameEnv batches are built on the GPU; device {self.device} is not usable "
                                      "(no CPU fallback)")
            if getattr(self, "_csr", None) is not None:
                self._store = ReplayStore.from_arrays(*self._csr, self.device)
            else:
                user_dict = self.base.train_user_dataset.user_dict
                ids = list(self.base.train_user_dataset.users) + list(self.base.test_user_dataset.users)
                self._store = ReplayStore(ids, user_dict, self.device)
            self._table = self.base.embeddings.to(self.device, torch.float32).contiguous()
        return self._store

    @property
    def table(self) -> torch.Tensor:
        self.store
        return self._table

    # ------------------------------------------------------------------ batches
    def collate_users(self, user_ids):
        """Batch of the given users, windows concatenated in the given order (prepare_batch_static_size)."""
        return self.collate_slots(self.store.slots(user_ids), user_ids)

    def collate_slots(self, slots, user_ids=None, rows_per_batch="env"):
        """The same for users given by their slots in the replay store (what the engine's sampler permutes)."""
        st = self.store
        slots = np.asarray(slots, dtype=np.int32)          # (the gather kernels read 32-bit slots)
        if user_ids is None:
            user_ids = slots
        sizes = st.lengths[slots]
        total = int(np.maximum(sizes - self.frame_size, 0).sum())
        cut = self.rows_per_batch if rows_per_batch == "env" else rows_per_batch
        rows = total if cut is None else min(cut, total)
        meta = {"users": torch.as_tensor(np.asarray(list(user_ids))), "sizes": torch.from_numpy(sizes.copy())}
        if self.embed_batch is utils.batch_tensor_embeddings:
            users_d = torch.from_numpy(slots).to(self.device)
            batch = utils.gather_frames(st.items, st.ratings, st.user_off, users_d, total, rows, self.frame_size, self._table,
                                        contiguous=self.contiguous)
            batch["meta"] = meta
            return batch
        # custom embed function: give it the windowed index batch on the device (generic torch path)
        f1 = self.frame_size + 1
        starts = np.concatenate([st_off + np.arange(max(L - self.frame_size, 0)) for st_off, L in
                                 zip(st.user_off.cpu().numpy()[slots], sizes)]) if len(slots) else np.zeros(0, np.int64)
        idx = torch.from_numpy(starts[:rows, None] + np.arange(f1)[None, :]).to(self.device)
        win = {"items": st.items[idx].long(), "ratings": st.ratings[idx], "sizes": meta["sizes"].to(self.device),
               "users": meta["users"]}
        return self.embed_batch(batch=win, item_embeddings_tensor=self._table, frame_size=self.frame_size)

    def collate_rows(self, seq_slots, skip0: int, row_start: int, rows: int):
        """`rows` consecutive rows, starting at global row `row_start`, of the concatenated windows of the users `seq_slots` (store
        slots; the first one with its first `skip0` windows left out): what a DENSE epoch's batch holds (fused.attach_sampler).  The
        users that overlap the range are collated whole -- `prepare_batch_static_size` semantics, `done` at each user's last
        window -- and the range is cut out of that."""
        st = self.store
        seq = np.asarray(seq_slots, dtype=np.int64)
        wins = np.maximum(st.lengths[seq].astype(np.int64) - self.frame_size, 0)
        wins[0] = max(int(wins[0]) - int(skip0), 0)
        cum = np.cumsum(wins)
        if row_start + rows > int(cum[-1]):
            raise IndexError("collate_rows: the sequence holds fewer rows")
        i0 = int(np.searchsorted(cum, row_start, side="right"))
        i1 = int(np.searchsorted(cum, row_start + rows - 1, side="right"))
        before = int(cum[i0 - 1]) if i0 > 0 else 0
        a = row_start - before + (int(skip0) if i0 == 0 else 0)       # offset inside the whole-user collate of seq[i0 .. i1]
        whole = self.collate_slots(seq[i0:i1 + 1].astype(np.int32), rows_per_batch=None)
        out = {k: (v[a:a + rows] if isinstance(v, torch.Tensor) else v) for k, v in whole.items()}
        return out

    def prepare_batch_wrapper(self, x):
        """collate_fn-compatible entry (env.py:241-248): x = list of UserDataset items."""
        return self.collate_users([b["users"] for b in x])

    def train_batch(self):
        """A fresh shuffled iterator per call, first batch of it (env.py:250-252)."""
        return next(iter(self.train_dataloader))

    def test_batch(self):
        return next(iter(self.test_dataloader))
