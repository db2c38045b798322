"""Exact batched nearest-item search on the GPU: the retrieval step that turns generated actions into item ids.

`FlatIndex(table, metric)` mirrors how the reference's demo uses faiss (`examples/streamlit_demo.py:190-204`:
IndexFlatL2 / IndexFlatIP / IndexFlatIP over L2-normalised rows) and `MilvusConnection.search`
(`recnn/data/db_con.py:45-56`): `search(queries, k)` returns `(distances[B, k], ids[B, k])`, best first.
SURVEY.md 8 row f2 ("next"); kernel in csrc/topk.hip.
"""
import ctypes as C

import torch

from . import _lib as L

METRICS = {"IP": 0, "L2": 1, "COS": 2}


class FlatIndex:
    def __init__(self, table: torch.Tensor, metric: str = "L2"):
        if metric not in METRICS:
            raise ValueError(f"metric must be one of {sorted(METRICS)}")
        if not table.is_cuda:
            raise L.RecnnHipError("FlatIndex: the item table must live on the GPU (no CPU fallback)")
        self.metric = metric
        self.table = table.detach().to(torch.float32).contiguous()
        self.n_items, self.dim = self.table.shape
        self.aux = None
        if metric != "IP":
            self.aux = torch.empty(self.n_items, dtype=torch.float32, device=table.device)
            L.call("recnn_topk_item_aux", L.ptr(self.table), self.n_items, self.dim, METRICS[metric], L.ptr(self.aux),
                   L.current_stream())

    @property
    def ntotal(self):
        return self.n_items

    def search(self, queries: torch.Tensor, k: int = 10):
        """(distances float32[B, k], ids int64[B, k]); L2 -> squared distances ascending, IP / COS -> scores descending."""
        q = queries.detach().to(self.table.device, torch.float32)
        if q.dim() == 1:
            q = q[None]
        if q.stride(-1) != 1 or q.stride(0) % 4 or q.data_ptr() % 16:
            q = q.contiguous()
        B = q.shape[0]
        dist = torch.empty(B, k, dtype=torch.float32, device=q.device)
        ids = torch.empty(B, k, dtype=torch.int64, device=q.device)
        nbytes = C.c_int64()
        L.call("recnn_topk_workspace_bytes", B, k, C.byref(nbytes))
        ws = torch.empty(max(int(nbytes.value), 16), dtype=torch.uint8, device=q.device)
        L.call("recnn_topk_search", L.ptr(q), q.stride(0), B, L.ptr(self.table), self.n_items, self.dim, METRICS[self.metric],
               L.ptr(self.aux), k, L.ptr(dist), L.ptr(ids), L.ptr(ws), L.current_stream())
        return dist, ids
