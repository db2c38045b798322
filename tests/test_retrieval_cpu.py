"""CPU: the retrieval oracle against scipy, the library the reference's `rank` helper calls
(examples/streamlit_demo.py:207-231: per-item `metric(embedding, gen_action)`, sorted ascending, first k)."""
import numpy as np
from scipy.spatial import distance

from oracle import retrieval_oracle as R


def test_oracle_matches_scipy_rank_semantics():
    rng = np.random.default_rng(0)
    table = rng.standard_normal((500, 128)).astype(np.float32)
    q = rng.standard_normal((7, 128)).astype(np.float32)
    d, i = R.topk(q, table, "L2", 10)
    ref = distance.cdist(q.astype(np.float64), table.astype(np.float64), "sqeuclidean")
    assert np.array_equal(i, np.argsort(ref, axis=1, kind="stable")[:, :10])
    assert np.allclose(d, np.sort(ref, axis=1)[:, :10], rtol=1e-9)
    # the demo's `rank` with scipy's euclidean metric orders items the same way
    eu = distance.cdist(q.astype(np.float64), table.astype(np.float64), "euclidean")
    assert np.array_equal(i, np.argsort(eu, axis=1, kind="stable")[:, :10])
    # cosine index of the demo: IP over normalised rows == 1 - scipy cosine distance up to the query norm
    dc, ic = R.topk(q, table, "COS", 10)
    cos = distance.cdist(q.astype(np.float64), table.astype(np.float64), "cosine")
    assert np.array_equal(ic, np.argsort(cos, axis=1, kind="stable")[:, :10])
    assert np.allclose(dc / np.linalg.norm(q.astype(np.float64), axis=1, keepdims=True), 1.0 - np.sort(cos, axis=1)[:, :10])
    di, ii = R.topk(q, table, "IP", 5)
    assert np.array_equal(ii, np.argsort(-(q.astype(np.float64) @ table.T.astype(np.float64)), axis=1, kind="stable")[:, :5])


def test_ties_go_to_the_smaller_id():
    table = np.zeros((6, 128), dtype=np.float32)
    table[[1, 4], 0] = 1.0
    q = np.zeros((1, 128), dtype=np.float32)
    q[0, 0] = 1.0
    _, i = R.topk(q, table, "IP", 3)
    assert i.tolist() == [[1, 4, 0]]
