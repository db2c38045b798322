"""CPU oracle for the device-side replay-store builder.

TEST INFRASTRUCTURE -- NOT A PRODUCT PATH.  Only `tests/` (and oracle/make_golden_etl.py) may import this module.

Restates recnn/data/dataset_functions.py:84-126 (`prepare_dataset`) as array operations:
  :100      rating -> 2 (rating - 2.5)
  :101      movieId -> key_to_id[movieId]
  :103-105  users = ids with more than frame_size ratings, by count descending
  :107      df.sort_values(by="timestamp")
  :110-121  groupby("userId"): per user the items / ratings columns in the sorted order
with ONE deliberate choice: rows that share (user, timestamp) are kept in INPUT order (a stable sort).  The reference's
sort is numpy's default, unstable argsort, so its order inside such tie groups is an accident of the host's numpy build
(introsort, or the AVX-512 sort); any order there is a correct output of the reference's algorithm.  Pinned against the
real reference by oracle/make_golden_etl.py: identical on a fixture without ties (tests/golden/etl_unique.npz), identical
up to the order inside tie groups on one with heavy ties (tests/golden/etl_ties.npz).
"""
import numpy as np


def csr_stable(user_ids, items_dense, ratings_raw, timestamps):
    """(users ascending, user_off int64[U+1], items int64[n], ratings float64[n]) ordered by (user, timestamp, input row)."""
    user_ids = np.asarray(user_ids, dtype=np.int64)
    n = len(user_ids)
    order = np.lexsort((np.arange(n), np.asarray(timestamps, dtype=np.int64), user_ids))     # last key is the primary one
    u = user_ids[order]
    first = np.flatnonzero(np.concatenate(([True], u[1:] != u[:-1]))) if n else np.zeros(0, dtype=np.int64)
    off = np.concatenate((first, [n])).astype(np.int64)
    return u[first], off, np.asarray(items_dense, dtype=np.int64)[order], 2.0 * (np.asarray(ratings_raw, dtype=np.float64)[order] - 2.5)


def same_up_to_tie_order(users, off, items, ratings, ref_items, ref_ratings, in_user, in_ts, in_items_dense):
    """True if (items, ratings) and (ref_items, ref_ratings) -- two row orders of the same CSR -- agree everywhere except
    for permutations INSIDE groups of rows sharing (user, timestamp).  The timestamps of the output rows are recovered from
    the inputs by sorting (user, timestamp) pairs, which both orders share."""
    in_user, in_ts = np.asarray(in_user, dtype=np.int64), np.asarray(in_ts, dtype=np.int64)
    o = np.lexsort((in_ts, in_user))
    ts_sorted, u_sorted = in_ts[o], in_user[o]
    if len(items) != len(ref_items):
        return False
    new_group = np.concatenate(([True], (ts_sorted[1:] != ts_sorted[:-1]) | (u_sorted[1:] != u_sorted[:-1])))
    gid = np.cumsum(new_group) - 1
    for a, b in ((items, ref_items), (ratings, ref_ratings)):
        ka = np.lexsort((np.asarray(a), gid))
        kb = np.lexsort((np.asarray(b), gid))
        if not np.array_equal(np.asarray(a)[ka], np.asarray(b)[kb]):
            return False
    # (item, rating) pairs must travel together
    pa = np.lexsort((ratings, items, gid))
    pb = np.lexsort((ref_ratings, ref_items, gid))
    return bool(np.array_equal(items[pa], ref_items[pb]) and np.array_equal(ratings[pa], ref_ratings[pb]))
