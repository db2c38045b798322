"""Ratings ETL: time the vectorised CSR builder against the reference's prepare_dataset on a
synthetic ML20M-shaped frame.  CPU only; needs the reference at /root/reference (build container).  Prints one JSON line.

  python tools/bench_etl.py [n_rows]
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import pandas

n_rows = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
n_users = max(100, int(138_493 * n_rows / 20_000_263))
n_items = 26_744
rng = np.random.default_rng(0)
keys = np.sort(rng.choice(np.arange(1, 200_000), size=n_items, replace=False))
frame = {"userId": rng.integers(1, n_users + 1, n_rows), "movieId": rng.choice(keys, n_rows),
         "rating": rng.integers(1, 11, n_rows) * 0.5, "timestamp": rng.integers(789_652_009, 1_427_784_002, n_rows)}
key_to_id = {int(k): i for i, k in enumerate(keys)}


class Base:
    pass


def run(mod):
    base = Base()
    base.key_to_id = dict(key_to_id)
    args = mod.DataFuncArgsMut(df=pandas.DataFrame(frame), base=base, users=None, user_dict=None)
    t0 = time.perf_counter()
    args, _ = mod.prepare_dataset(args, mod.DataFuncKwargs(frame_size=10))
    return time.perf_counter() - t0, args


from recnn_amd.data import dataset_functions as ours
t_ours, a = run(ours)
out = {"metric": "ratings ETL: prepare_dataset seconds", "n_rows": n_rows, "n_users": n_users, "ours_s": t_ours, "host_threads": os.cpu_count()}
ref_root = os.environ.get("RECNN_REFERENCE", "/root/reference")
if os.path.isdir(ref_root):
    sys.path.insert(0, ref_root)
    sys.path.insert(0, os.path.join(ROOT, "oracle", "_stubs"))
    for m in [k for k in sys.modules if k == "recnn" or k.startswith("recnn.")]:
        del sys.modules[m]
    from recnn.data import dataset_functions as refmod
    t_ref, b = run(refmod)
    out["reference_s"] = t_ref
    out["speedup"] = t_ref / t_ours
    u = next(iter(a.user_dict))
    same = all(np.array_equal(a.user_dict[k]["items"], b.user_dict[k]["items"]) and
               np.array_equal(a.user_dict[k]["ratings"], b.user_dict[k]["ratings"]) for k in list(a.user_dict)[:2000])
    out["identical_first_2000_users"] = bool(same and list(a.users) == list(b.users))
print(json.dumps(out))
