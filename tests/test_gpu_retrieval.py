"""GPU parity of the fused scoring + top-K kernel (csrc/topk.hip) against the float64 oracle.

Ids must match exactly wherever the oracle's gap between consecutive ranked scores exceeds fp32 round-off of the
score; inside such near-ties the returned id must still be one of the tied candidates and its true score must equal
the oracle's at that rank within round-off.  Reported distances within 1e-5 relative."""
import numpy as np
import pytest
import torch

from oracle import retrieval_oracle as R

pytestmark = pytest.mark.gpu


def _check(dist, ids, q, table, metric, k):
    d_ref, i_ref = R.topk(q, table, metric, k)
    s_all = R.scores(q, table, metric)
    ids = ids.cpu().numpy()
    dist = dist.cpu().numpy().astype(np.float64)
    scale = np.abs(s_all).max()
    tol = 4e-6 * scale
    got_true = np.take_along_axis(s_all, ids, 1)
    assert np.abs(got_true - d_ref).max() <= tol                 # right score at every rank (ids may swap inside a tie)
    assert np.abs(dist - got_true).max() <= 1e-5 * scale         # reported distance = true score of the reported id
    for b in range(ids.shape[0]):
        assert len(set(ids[b].tolist())) == k
    gap_ok = np.ones_like(i_ref, dtype=bool)                      # ranks whose neighbours are clearly separated
    gap = np.abs(np.diff(d_ref, axis=1))
    gap_ok[:, 1:] &= gap > 4 * tol
    gap_ok[:, :-1] &= gap > 4 * tol
    nxt = np.abs(np.sort(s_all if metric == "L2" else -s_all, axis=1)[:, k] - (d_ref[:, -1] if metric == "L2" else -d_ref[:, -1]))
    gap_ok[:, -1] &= nxt > 4 * tol
    assert np.array_equal(ids[gap_ok], i_ref[gap_ok])
    return gap_ok.mean()


@pytest.mark.parametrize("metric", ["L2", "IP", "COS"])
@pytest.mark.parametrize("B,N,k", [(2048, 26744, 10), (100, 5000, 64), (1, 300, 1), (65, 100000, 20)])
def test_topk_matches_oracle(cuda, metric, B, N, k):
    from recnn_amd.retrieval import FlatIndex
    rng = np.random.default_rng(B + N + k)
    table = rng.standard_normal((N, 128)).astype(np.float32)
    q = (rng.standard_normal((B, 128)) * 0.7).astype(np.float32)
    idx = FlatIndex(torch.from_numpy(table).to(cuda), metric)
    assert idx.ntotal == N
    d, i = idx.search(torch.from_numpy(q).to(cuda), k)
    torch.cuda.synchronize()
    frac = _check(d, i, q, table, metric, k)
    assert frac > 0.95                                             # the exact-id comparison is not vacuous


def test_topk_ties_and_duplicates(cuda):
    from recnn_amd.retrieval import FlatIndex
    table = np.zeros((700, 128), dtype=np.float32)
    table[:, 0] = np.arange(700) % 7                               # 100 exact duplicates of each of 7 rows
    q = np.zeros((3, 128), dtype=np.float32)
    q[:, 0] = [1.0, -1.0, 0.5]
    for metric in ("IP", "L2"):
        d, i = FlatIndex(torch.from_numpy(table).to(cuda), metric).search(torch.from_numpy(q).to(cuda), 12)
        d_ref, i_ref = R.topk(q, table, metric, 12)
        assert np.array_equal(i.cpu().numpy(), i_ref)              # ties resolved towards the smaller id, as the oracle
        assert np.allclose(d.cpu().numpy(), d_ref, atol=1e-6)


def test_actor_output_to_items_roundtrip(cuda):
    """The demo's flow (streamlit_demo.py:190-231): generated action -> nearest items; a table row retrieves itself."""
    from recnn_amd.retrieval import FlatIndex
    rng = np.random.default_rng(1)
    table = torch.from_numpy(rng.standard_normal((26744, 128)).astype(np.float32)).to(cuda)
    pick = torch.tensor([0, 17, 26743, 5000], device=cuda)
    d, i = FlatIndex(table, "L2").search(table[pick], 5)
    assert torch.equal(i[:, 0], pick) and float(d[:, 0].abs().max()) < 1e-3
