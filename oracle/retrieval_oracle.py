"""CPU oracle of the exact top-K item search (TEST INFRASTRUCTURE ONLY).

Restates what the reference's retrieval call sites compute:
  examples/streamlit_demo.py:190-204  faiss IndexFlatL2 (squared L2, ascending), IndexFlatIP (inner product, descending),
                                      IndexFlatIP over L2-normalised table rows ("COS": q . t/|t|, descending)
  examples/streamlit_demo.py:207-231  `rank(gen_action, metric, k)`: per-item scipy distance, sorted ascending
  recnn/data/db_con.py:45-56          MilvusConnection.search(vecs, topk) -> ids / distances (L2 collection)
faiss and pymilvus are third-party services that are not vendored (`requirements.txt`); their flat search is exact
brute force, which this file restates in float64 numpy.  Pinned against scipy (the library `rank` calls) by
tests/test_retrieval_cpu.py.  Ties are broken towards the smaller item id (documented choice; faiss leaves it open).
"""
import numpy as np


def scores(queries: np.ndarray, table: np.ndarray, metric: str) -> np.ndarray:
    """float64 [B, N]: the quantity that is reported (L2: squared distance; IP, COS: similarity)."""
    q = np.asarray(queries, dtype=np.float64)
    t = np.asarray(table, dtype=np.float64)
    if metric == "IP":
        return q @ t.T
    if metric == "COS":
        return q @ (t / np.linalg.norm(t, axis=1, keepdims=True)).T
    if metric == "L2":
        return (q * q).sum(1)[:, None] - 2.0 * (q @ t.T) + (t * t).sum(1)[None, :]
    raise ValueError(metric)


def topk(queries, table, metric: str, k: int):
    """(dist float64[B,k], ids int64[B,k]) best first."""
    s = scores(queries, table, metric)
    key = s if metric == "L2" else -s
    n = s.shape[1]
    order = np.lexsort((np.broadcast_to(np.arange(n), s.shape), key), axis=1)[:, :k]     # by key, then by id
    return np.take_along_axis(s, order, 1), order.astype(np.int64)
