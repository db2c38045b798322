#!/usr/bin/env python3
"""Generate `tests/golden/etl_ties.npz` from the REAL reference's ratings ETL.

TEST INFRASTRUCTURE.  Runs only in the build container (reference mounted at /root/reference).  Drives the reference's
own `recnn.data.dataset_functions.prepare_dataset` (dataset_functions.py:84-126) on a seeded synthetic ratings frame
with many EQUAL timestamps (the order of such rows is decided by pandas' unstable sort, which the vectorised
`recnn_amd.data.dataset_functions.csr_from_ratings` has to reproduce) and stores input + output.

Usage:  python oracle/make_golden_etl.py        (from the repo root)
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("RECNN_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(HERE, "_stubs"))

import numpy as np  # noqa: E402
import pandas  # noqa: E402

import recnn as ref  # noqa: E402  (the reference)

assert ref.__file__.startswith(REF), ref.__file__
from recnn.data import dataset_functions as RF  # noqa: E402

rng = np.random.default_rng(7)
n_rows, n_users, n_items = 6000, 90, 300
keys = np.sort(rng.choice(np.arange(1000, 5000), size=n_items, replace=False))
df = pandas.DataFrame({
    "userId": rng.integers(1, n_users + 1, n_rows),
    "movieId": rng.choice(keys, n_rows),
    "rating": rng.integers(1, 11, n_rows) * 0.5,
    "timestamp": rng.integers(0, 400, n_rows),          # ~15 rows per timestamp value: plenty of ties inside a user
})
inp = {c: df[c].to_numpy().copy() for c in df.columns}


class Base:
    pass


base = Base()
base.key_to_id = {int(k): i for i, k in enumerate(keys)}
args = RF.DataFuncArgsMut(df=df.copy(), base=base, users=None, user_dict=None)
args, _ = RF.prepare_dataset(args, RF.DataFuncKwargs(frame_size=10))
users = np.asarray(list(args.users), dtype=np.int64)
uids = np.asarray(sorted(args.user_dict.keys()), dtype=np.int64)
off = np.zeros(len(uids) + 1, dtype=np.int64)
off[1:] = np.cumsum([len(args.user_dict[u]["items"]) for u in uids])
items = np.concatenate([np.asarray(args.user_dict[u]["items"], dtype=np.int64) for u in uids])
ratings = np.concatenate([np.asarray(args.user_dict[u]["ratings"], dtype=np.float64) for u in uids])
out = os.path.join(ROOT, "tests", "golden", "etl_ties.npz")
np.savez_compressed(out, keys=keys, users_filtered=users, uids=uids, user_off=off, items=items, ratings=ratings,
                    **{"in_" + k: v for k, v in inp.items()})
print("wrote", out, "rows", n_rows, "users", len(uids), "eligible", len(users))
