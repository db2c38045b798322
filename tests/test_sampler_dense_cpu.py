"""Host logic of the DENSE sampler (recnn_amd/nn/fused.py dense_epoch): the fixed-row batches of an epoch are
consecutive cuts of the concatenated windows of the shuffled users -- every (user, window) exactly once, like the reference's
whole-user batches (recnn/data/utils.py:161-187) -- and what an epoch leaves over is carried into the next one."""
import numpy as np

from recnn_amd.nn.fused import dense_epoch

FRAME = 10


def plan_host(seq, skip0, user_off, lengths, frame, rows, n_e):
    """numpy restatement of recnn_frame_plan_dense for the first n_e * rows rows: (store slot, window index, done) per row."""
    out = []
    for i, u in enumerate(seq):
        w = max(int(lengths[u]) - frame, 0)
        t0 = skip0 if i == 0 else 0
        for t in range(t0, w):
            out.append((int(u), t, int(t == w - 1)))
            if len(out) == n_e * rows:
                return out
    return out


def test_every_window_is_visited_exactly_once_per_epoch():
    rng = np.random.default_rng(0)
    n_users, rows = 57, 64
    lengths = rng.integers(5, 90, size=n_users)           # some users have no window at all (L <= frame)
    lengths[3] = FRAME                                    # exactly zero windows
    user_off = np.concatenate([[0], np.cumsum(lengths)])
    train = np.arange(n_users, dtype=np.int32)
    total = int(np.maximum(lengths - FRAME, 0).sum())
    visits = {}
    carry = (np.zeros(0, np.int32), 0)
    n_epochs = 4
    batches = 0
    for e in range(n_epochs):
        perm = rng.permutation(train)
        seq, skip0, n_e, carry = dense_epoch(carry, perm, lengths, FRAME, rows)
        assert n_e in (total // rows, total // rows + 1)
        rows_e = plan_host(seq, skip0, user_off, lengths, FRAME, rows, n_e)
        assert len(rows_e) == n_e * rows
        for u, t, done in rows_e:
            visits[(u, t)] = visits.get((u, t), 0) + 1
            assert done == int(t == lengths[u] - FRAME - 1)             # `done` marks the user's last window wherever it falls
        batches += n_e
    # what the last epoch left over is still owed: with it, every (user, window) has been seen once per epoch
    cs, sk = carry
    for i, u in enumerate(cs):
        for t in range(sk if i == 0 else 0, max(int(lengths[u]) - FRAME, 0)):
            visits[(int(u), t)] = visits.get((int(u), t), 0) + 1
    want = {(u, t) for u in range(n_users) for t in range(max(int(lengths[u]) - FRAME, 0))}
    assert set(visits) == want
    assert all(v == n_epochs for v in visits.values()), sorted(set(visits.values()))
    leftover = sum(max(int(lengths[u]) - FRAME, 0) for u in cs) - sk
    assert 0 <= leftover < rows and batches * rows + leftover == n_epochs * total


def test_epoch_that_ends_exactly_on_a_user_boundary_carries_nothing():
    lengths = np.array([FRAME + 8, FRAME + 8, FRAME + 16])
    seq, skip0, n_e, carry = dense_epoch((np.zeros(0, np.int32), 0), np.array([2, 0, 1], np.int32), lengths, FRAME, 16)
    assert n_e == 2 and skip0 == 0 and len(carry[0]) == 0 and carry[1] == 0
    seq, skip0, n_e, carry = dense_epoch((np.zeros(0, np.int32), 0), np.array([0, 2, 1], np.int32), lengths, FRAME, 12)
    assert n_e == 2 and list(carry[0]) == [1] and carry[1] == 0            # 24 rows used: user 0 (8) + all 16 windows of user 2
    seq2, skip2, n2, carry2 = dense_epoch(carry, np.array([1, 0, 2], np.int32), lengths, FRAME, 12)
    assert list(seq2) == [1, 1, 0, 2] and skip2 == 0 and n2 == (8 + 32) // 12
    assert list(carry2[0]) == [2] and carry2[1] == 12                      # 36 rows used: 8 + 8 + 8 + 12 of user 2's 16 windows
    seq3, skip3, n3, carry3 = dense_epoch(carry2, np.array([0, 1, 2], np.int32), lengths, FRAME, 12)
    assert list(seq3) == [2, 0, 1, 2] and skip3 == 12 and n3 == (4 + 32) // 12 and len(carry3[0]) == 0
