#!/usr/bin/env python3
"""Generate `tests/golden/etl_ties.npz` / `etl_unique.npz` from the REAL reference's ratings ETL and pin
`oracle/etl_oracle.py` (the stable-order restatement the device builder is checked against) on both.

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
sys.path.append(ROOT)     # after the reference: `import recnn` must find /root/reference, not the repository's `recnn` shim

import numpy as np  # noqa: E402
import pandas  # noqa: E402

import recnn as ref  # noqa: E402  (the reference)

assert ref.__file__.startswith(REF), ref.__file__
from recnn.data import dataset_functions as RF  # noqa: E402

class Base:
    pass


def run(name, seed, n_rows, n_users, n_items, unique_ts):
    rng = np.random.default_rng(seed)
    keys = np.sort(rng.choice(np.arange(1000, 5000), size=n_items, replace=False))
    uid = rng.integers(1, n_users + 1, n_rows)
    mid = rng.choice(keys, n_rows)
    rat = rng.integers(1, 11, n_rows) * 0.5
    if unique_ts:      # no two rows share a timestamp: the order is fully determined, any correct sort reproduces it
        ts = rng.permutation(n_rows * 3)[:n_rows] + 1_000_000_000
    else:              # ~15 rows per timestamp value: plenty of ties inside a user (pandas' unstable sort orders them)
        ts = rng.integers(0, 400, n_rows)
    df = pandas.DataFrame({"userId": uid, "movieId": mid, "rating": rat, "timestamp": ts})
    inp = {c: df[c].to_numpy().copy() for c in df.columns}
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
    # pin the stable-order oracle (oracle/etl_oracle.py) against the reference run
    from oracle import etl_oracle as E
    dense = np.asarray([base.key_to_id[int(k)] for k in inp["movieId"]], dtype=np.int64)
    o_users, o_off, o_items, o_ratings = E.csr_stable(inp["userId"], dense, inp["rating"], inp["timestamp"])
    assert np.array_equal(o_users, uids) and np.array_equal(o_off, off)
    if unique_ts:
        assert np.array_equal(o_items, items) and np.array_equal(o_ratings, ratings)
    else:
        assert E.same_up_to_tie_order(o_users, o_off, o_items, o_ratings, items, ratings, inp["userId"], inp["timestamp"], dense)
    out = os.path.join(ROOT, "tests", "golden", name + ".npz")
    np.savez_compressed(out, keys=keys, users_filtered=users, uids=uids, user_off=off, items=items, ratings=ratings,
                        **{"in_" + k: v for k, v in inp.items()})
    print("wrote", out, "rows", n_rows, "users", len(uids), "eligible", len(users), "| stable-order oracle:",
          "identical" if unique_ts else "identical up to the order inside (user, timestamp) tie groups")


run("etl_ties", 7, 6000, 90, 300, unique_ts=False)
run("etl_unique", 8, 6000, 90, 300, unique_ts=True)
