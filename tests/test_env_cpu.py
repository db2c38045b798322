"""CPU tests of the host-side FrameEnv logic (csv/pickle ETL contract, cache, loaders): no GPU work involved.
Reference behaviour: recnn/data/env.py:81-187, recnn/data/dataset_functions.py:84-126, recnn/data/utils.py:203-214."""
import os
import pickle

import numpy as np
import pandas as pd
import pytest
import torch

import recnn_amd
from recnn_amd.data import env as E


@pytest.fixture()
def toy_dir(tmp_path):
    """The toy-dataset recipe of the reference docs (docs/source/examples/your_data.rst:57-98), smaller."""
    rng = np.random.default_rng(0)
    n_users, n_items, n_rows = 40, 60, 1500
    df = pd.DataFrame({"userId": rng.integers(1, n_users + 1, n_rows), "movieId": rng.integers(1000, 1000 + n_items, n_rows),
                       "rating": rng.integers(1, 11, n_rows) * 0.5, "timestamp": rng.permutation(n_rows)})
    df.to_csv(tmp_path / "ratings.csv", index=False)
    emb = {int(k): torch.randn(16) for k in range(1000, 1000 + n_items)}
    with open(tmp_path / "emb.pkl", "wb") as f:
        pickle.dump(emb, f)
    return tmp_path, df, emb


def test_frame_env_from_csv_builds_the_reference_data_contract(toy_dir):
    tmp, df, emb = toy_dir
    path = E.DataPath(base=str(tmp) + "/", ratings="ratings.csv", embeddings="emb.pkl", cache="cache.pkl", use_cache=True)
    env = E.FrameEnv(path, frame_size=10, batch_size=5)
    base = env.base
    assert base.embeddings.shape == (60, 16) and len(base.key_to_id) == 60
    assert base.key_to_id[1000] == 0 and base.id_to_key[59] == 1059                      # sorted keys -> dense ids
    assert torch.equal(base.embeddings[base.key_to_id[1017]], emb[1017])
    ud = base.train_user_dataset.user_dict
    counts = df.groupby("userId").size()
    eligible = set(counts[counts > 10].index)
    users = set(base.train_user_dataset.users) | set(base.test_user_dataset.users)
    assert users <= eligible and len(users) == len(eligible) - 2                          # env.py:178 drops the 2 longest
    assert len(base.test_user_dataset) == int(np.ceil(0.05 * len(eligible)))             # test_size = 0.05
    u = next(iter(users))
    rows = df[df.userId == u].sort_values("timestamp")
    assert np.array_equal(ud[u]["items"], rows.movieId.map(base.key_to_id).values)       # time order, dense ids
    assert np.allclose(ud[u]["ratings"], 2 * (rows.rating.values - 2.5))                 # [0.5,5] -> [-4,5]
    item = base.train_user_dataset[0]
    assert set(item) == {"items", "rates", "sizes", "users"} and item["sizes"] == len(item["items"])
    lens = [len(ud[x]["items"]) for x in base.train_user_dataset.users]
    assert lens == sorted(lens, reverse=True)                                            # sort_users_itemwise
    assert len(env.train_dataloader) == -(-len(base.train_user_dataset) // 5)
    assert env.frame_size == 10 and env.batch_size == 5 and env.num_workers == 1
    # cache round trip: second construction loads the pickle and yields the same split
    assert os.path.isfile(tmp / "cache.pkl")
    env2 = E.FrameEnv(path, frame_size=10, batch_size=5)
    assert list(env2.base.train_user_dataset.users) == list(base.train_user_dataset.users)
    # the pickle unpickles under the reference's module name too
    recnn_amd.install_as("recnn")
    import recnn.data.env as alias
    assert alias.EnvBase is E.EnvBase


def test_pipeline_helpers(toy_dir):
    tmp, df, emb = toy_dir
    from recnn_amd.data import dataset_functions as D
    kw = D.DataFuncKwargs(frame_size=10)
    with pytest.raises(AttributeError):
        kw.get("reduce_items_to")
    kw.set("reduce_items_to", 20)
    base = E.EnvBase()
    base.embeddings, base.key_to_id, base.id_to_key = recnn_amd.data.make_items_tensor(emb)
    args = D.DataFuncArgsMut(df=df.copy(), base=base, users=None, user_dict=None)
    D.build_data_pipeline([D.truncate_dataset, D.prepare_dataset], kw, args)
    assert args.base.embeddings.shape[0] == 20 and len(args.base.key_to_id) == 20
    assert all(it.max() < 20 for it in (d["items"] for d in args.user_dict.values()))


def test_rolling_window_and_get_base_batch_shapes():
    a = np.arange(7)
    w = recnn_amd.data.rolling_window(a, 3)
    assert w.shape == (5, 3) and w[4].tolist() == [4, 5, 6]
    b = {"state": torch.zeros(4, 6), "action": torch.zeros(4, 2), "reward": torch.zeros(4), "next_state": torch.zeros(4, 6),
         "done": torch.ones(4)}
    out = recnn_amd.data.get_base_batch(b, device=torch.device("cpu"))
    assert [tuple(t.shape) for t in out] == [(4, 6), (4, 2), (4, 1), (4, 6), (4, 1)]
    assert len(recnn_amd.data.get_base_batch(b, device=torch.device("cpu"), done=False)) == 4


def test_vectorised_etl_matches_reference_fixture_including_timestamp_ties(golden_dir):
    """csr_from_ratings / prepare_dataset against the REAL reference's prepare_dataset output
    (tests/golden/etl_ties.npz, oracle/make_golden_etl.py): same users, same per-user item and rating sequences,
    bit for bit -- also for rows with equal timestamps, whose order pandas' unstable sort decides."""
    import pandas
    from recnn_amd.data import dataset_functions as F
    g = np.load(os.path.join(golden_dir, "etl_ties.npz"))
    key_to_id = {int(k): i for i, k in enumerate(g["keys"])}
    dense = np.asarray([key_to_id[int(k)] for k in g["in_movieId"]], dtype=np.int64)
    users, off, items, ratings = F.csr_from_ratings(g["in_userId"], dense, 2.0 * (g["in_rating"] - 2.5), g["in_timestamp"])
    assert np.array_equal(users, g["uids"]) and np.array_equal(off, g["user_off"])
    assert np.array_equal(items, g["items"]) and np.array_equal(ratings, g["ratings"])

    class Base:
        pass
    base = Base()
    base.key_to_id = key_to_id
    df = pandas.DataFrame({"userId": g["in_userId"], "movieId": g["in_movieId"], "rating": g["in_rating"],
                           "timestamp": g["in_timestamp"]})
    args = F.DataFuncArgsMut(df=df, base=base, users=None, user_dict=None)
    args, _ = F.prepare_dataset(args, F.DataFuncKwargs(frame_size=10))
    assert list(args.users) == list(g["users_filtered"])            # > frame_size ratings, longest history first
    for i, u in enumerate(g["uids"]):
        a, b = g["user_off"][i], g["user_off"][i + 1]
        assert np.array_equal(args.user_dict[int(u)]["items"], g["items"][a:b])
        assert np.array_equal(args.user_dict[int(u)]["ratings"], g["ratings"][a:b])


def test_csr_from_ratings_edge_cases():
    from recnn_amd.data import dataset_functions as F
    e = np.zeros(0, dtype=np.int64)
    users, off, items, ratings = F.csr_from_ratings(e, e, e.astype(np.float64), e)
    assert len(users) == 0 and off.tolist() == [0] and len(items) == 0
    # one user, already in time order; ids above 2^16 take the two-pass radix path
    u = np.full(5, 70000, dtype=np.int64)
    users, off, items, ratings = F.csr_from_ratings(u, np.arange(5), np.arange(5) * 0.5, np.arange(5))
    assert users.tolist() == [70000] and off.tolist() == [0, 5] and items.tolist() == [0, 1, 2, 3, 4]
    # interleaved users, reversed time
    u = np.array([3, 1, 3, 1, 2], dtype=np.int64)
    users, off, items, ratings = F.csr_from_ratings(u, np.array([10, 11, 12, 13, 14]), np.ones(5), np.array([5, 4, 3, 2, 1]))
    assert users.tolist() == [1, 2, 3] and off.tolist() == [0, 2, 3, 5] and items.tolist() == [13, 11, 14, 12, 10]
    assert F._stable_argsort_ids(np.array([-1, 5, 2])).tolist() == [0, 2, 1]          # negative ids: generic stable sort


def test_stable_order_etl_oracle_against_reference_fixtures(golden_dir):
    """oracle/etl_oracle.py (the order the device builder produces: ties inside (user, timestamp) in input order) against
    the REAL reference's prepare_dataset: identical without ties, identical up to the order inside tie groups with them."""
    from oracle import etl_oracle as E
    for name, exact in (("etl_unique.npz", True), ("etl_ties.npz", False)):
        g = np.load(os.path.join(golden_dir, name))
        key_to_id = {int(k): i for i, k in enumerate(g["keys"])}
        dense = np.asarray([key_to_id[int(k)] for k in g["in_movieId"]], dtype=np.int64)
        users, off, items, ratings = E.csr_stable(g["in_userId"], dense, g["in_rating"], g["in_timestamp"])
        assert np.array_equal(users, g["uids"]) and np.array_equal(off, g["user_off"])
        if exact:
            assert np.array_equal(items, g["items"]) and np.array_equal(ratings, g["ratings"])
        else:
            assert not np.array_equal(items, g["items"])          # the fixture does exercise the tie order
            assert E.same_up_to_tie_order(users, off, items, ratings, g["items"], g["ratings"], g["in_userId"],
                                          g["in_timestamp"], dense)
            swapped = g["items"].copy()
            swapped[[0, len(swapped) - 1]] = swapped[[len(swapped) - 1, 0]]      # a swap ACROSS groups must be caught
            assert not E.same_up_to_tie_order(users, off, items, ratings, swapped, g["ratings"], g["in_userId"],
                                              g["in_timestamp"], dense)
