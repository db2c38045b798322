"""GPU parity of the replay-store builder: recnn_csr_build / csr_from_ratings_device /
prepare_dataset_device against the stable-order oracle (oracle/etl_oracle.py) and the fixtures of the real reference."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _fixture(golden_dir, name):
    g = np.load(os.path.join(golden_dir, name))
    key_to_id = {int(k): i for i, k in enumerate(g["keys"])}
    dense = np.asarray([key_to_id[int(k)] for k in g["in_movieId"]], dtype=np.int64)
    return g, key_to_id, dense


def _check(got, want):
    for a, b, name in zip(got, want, ("users", "user_off", "items", "ratings")):
        assert a.dtype == b.dtype and np.array_equal(a, b), name


@pytest.mark.parametrize("name", ["etl_unique.npz", "etl_ties.npz"])
def test_device_builder_on_the_reference_fixtures(cuda, golden_dir, name):
    """bit-identical to the real reference's prepare_dataset where (user, timestamp) pairs are unique; with ties identical
    to the stable-order oracle, i.e. to the reference up to the order inside tie groups."""
    from oracle import etl_oracle as E
    from recnn_amd.data import dataset_functions as F
    g, key_to_id, dense = _fixture(golden_dir, name)
    got = F.csr_from_ratings_device(g["in_userId"], g["in_movieId"], g["in_rating"], g["in_timestamp"], key_to_id=key_to_id,
                                    want_mapped_rows=True)
    assert np.array_equal(got[4], dense)                                  # the mapped id column in input order
    _check(got[:4], E.csr_stable(g["in_userId"], dense, g["in_rating"], g["in_timestamp"]))
    assert np.array_equal(got[0], g["uids"]) and np.array_equal(got[1], g["user_off"])
    if name == "etl_unique.npz":
        assert np.array_equal(got[2], g["items"]) and np.array_equal(got[3], g["ratings"])
    else:
        assert E.same_up_to_tie_order(got[0], got[1], got[2], got[3], g["items"], g["ratings"], g["in_userId"], g["in_timestamp"],
                                      dense)


@pytest.mark.parametrize("n,n_users,ts_hi,u_lo", [(1, 1, 5, 7), (63, 3, 4, 0), (4097, 50, 10 ** 9, 1), (300_000, 5000, 10 ** 9, 1),
                                                  (2_000_003, 140_000, 6 * 10 ** 8, 1), (50_000, 300, 50, -40),
                                                  (20_000, 17, 2 ** 40, 2 ** 33)])
def test_device_builder_equals_the_stable_oracle(cuda, n, n_users, ts_hi, u_lo):
    """sizes around the tile / wave boundaries, heavy ties, negative and > 2^32 user ids, 41-bit timestamps, 2 M rows at
    ML20M's shape (18 + 30 key bits: 6 radix passes)."""
    from oracle import etl_oracle as E
    from recnn_amd.data import dataset_functions as F
    rng = np.random.default_rng(n)
    users = rng.integers(u_lo, u_lo + n_users, n)
    items = rng.integers(0, 26_744, n)
    ratings = rng.integers(1, 11, n) * 0.5
    ts = rng.integers(0, ts_hi, n)
    got = F.csr_from_ratings_device(users, items, ratings, ts)
    _check(got, E.csr_stable(users, items, ratings, ts))


def test_device_builder_reports_unmapped_keys_and_prepare_falls_back(cuda, golden_dir):
    import pandas
    from recnn_amd.data import dataset_functions as F
    g, key_to_id, dense = _fixture(golden_dir, "etl_unique.npz")
    short = dict(list(key_to_id.items())[:-1])                            # one key missing -> host path semantics
    assert F.csr_from_ratings_device(g["in_userId"], g["in_movieId"], g["in_rating"], g["in_timestamp"], key_to_id=short) is None

    class Base:
        pass
    base = Base()
    base.key_to_id = key_to_id
    df = pandas.DataFrame({"userId": g["in_userId"], "movieId": g["in_movieId"], "rating": g["in_rating"],
                           "timestamp": g["in_timestamp"]})
    args = F.DataFuncArgsMut(df=df, base=base, users=None, user_dict=None)
    args, _ = F.prepare_dataset_device(args, F.DataFuncKwargs(frame_size=10))
    assert list(args.users) == list(g["users_filtered"])                 # > frame_size ratings, longest history first
    for i, u in enumerate(g["uids"]):
        a, b = g["user_off"][i], g["user_off"][i + 1]
        assert np.array_equal(args.user_dict[int(u)]["items"], g["items"][a:b])
        assert np.array_equal(args.user_dict[int(u)]["ratings"], g["ratings"][a:b])
    assert np.array_equal(args.df["movieId"].to_numpy(), dense)
    assert np.array_equal(args.df["rating"].to_numpy(), 2.0 * (g["in_rating"] - 2.5))


def test_frame_env_builds_through_the_device_etl(cuda, tmp_path):
    """FrameEnv(DataPath(csv, embeddings), prepare_dataset=prepare_dataset_device): batches equal the host-ETL env's."""
    import pickle
    import pandas
    import torch
    from recnn_amd.data import env as ENV
    from recnn_amd.data import dataset_functions as F
    rng = np.random.default_rng(3)
    n_items, n = 60, 3000
    keys = np.sort(rng.choice(np.arange(100, 900), n_items, replace=False))
    emb = {int(k): torch.randn(16, generator=torch.Generator().manual_seed(int(k))) for k in keys}
    pickle.dump(emb, open(tmp_path / "emb.pkl", "wb"))
    pandas.DataFrame({"userId": rng.integers(1, 40, n), "movieId": rng.choice(keys, n), "rating": rng.integers(1, 11, n) * 0.5,
                      "timestamp": rng.permutation(n) + 10 ** 9}).to_csv(tmp_path / "ratings.csv", index=False)
    envs = []
    for prep in (F.prepare_dataset, F.prepare_dataset_device):
        np.random.seed(0)                                   # sklearn's train_test_split draws from numpy's global state
        path = ENV.DataPath(base=str(tmp_path) + "/", ratings="ratings.csv", embeddings="emb.pkl", use_cache=False)
        envs.append(ENV.FrameEnv(path, frame_size=5, batch_size=8, num_workers=0, prepare_dataset=prep))
    a, b = (e.base.train_user_dataset for e in envs)
    assert list(a.users) == list(b.users) and len(a.users) > 5
    for u in a.user_dict:
        assert np.array_equal(a.user_dict[u]["items"], b.user_dict[u]["items"])
        assert np.array_equal(a.user_dict[u]["ratings"], b.user_dict[u]["ratings"])
    torch.manual_seed(0)
    ba = envs[0].train_batch()
    torch.manual_seed(0)
    bb = envs[1].train_batch()
    for k in ("state", "action", "reward", "next_state", "done"):
        assert torch.equal(ba[k], bb[k]), k
