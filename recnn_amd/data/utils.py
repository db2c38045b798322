"""Batch construction helpers (reference: recnn/data/utils.py).

The collate of the reference (rolling_window + concatenate + emb[items] + cat + done scatter,
utils.py:7-10, :51-81, :161-187) is ONE HIP kernel here (`recnn_frame_gather`), fed by the
device-resident CSR store of `recnn_amd.data.store`.  These functions keep the reference's call
shapes so that user code (custom `embed_batch` callables, direct `get_base_batch` calls) still works.
"""
import numpy as np
import torch

from .. import _lib as L
from .pandas_backend import pd

__all__ = ["rolling_window", "get_irsu", "batch_tensor_embeddings", "batch_contstate_discaction", "prepare_batch_static_size", "make_items_tensor",
           "sort_users_itemwise", "get_base_batch", "packed_ld", "FrameBatch"]


def rolling_window(a, window):
    """Sliding windows of a 1-D array as a strided view (utils.py:7-10).  Host helper for user code;
    the engine never materialises windows."""
    return np.lib.stride_tricks.sliding_window_view(a, window)


def get_irsu(batch):
    return batch["items"], batch["ratings"], batch["sizes"], batch["users"]


def packed_ld(frame_size: int, emb_dim: int) -> int:
    """Row stride (floats) of the packed batch rows [action | state | 0-pad] the MFMA GEMMs read:
    the 16-byte aligned, zero-padded layout shared with the C engine (engine.hip setup_dims)."""
    state = frame_size * emb_dim + frame_size
    r128 = lambda x: (x + 127) // 128 * 128
    return max(r128(emb_dim + r128(state)), r128(emb_dim + state))


class FrameBatch(dict):
    """The SARS' dict the reference's collate returns (keys state, action, reward, next_state, done, meta),
    whose tensors are views into packed device rows.  `update()` binds those rows directly (no copy)."""
    packed = None   # (xs, xn, reward, done, rows, ld)


def _require_cuda(t, what):
    if not (isinstance(t, torch.Tensor) and t.is_cuda):
        raise L.RecnnHipError(f"{what}: needs a GPU tensor (recnn_amd has no CPU fallback); got "
                              f"{getattr(t, 'device', type(t))}")


def gather_frames(items, ratings, user_off, batch_users, row_total, rows, frame_size, table, contiguous=False,
                  row_off=None, xs=None, xn=None, reward=None, done=None):
    """plan + gather on the current stream.  Returns a FrameBatch (without `meta`)."""
    dev = table.device
    E = table.shape[1]
    S = frame_size * E + frame_size
    n_users = batch_users.numel()
    stream = L.current_stream()
    if n_users > 1024:      # big user lists: explicit plan launch; otherwise the gather kernel plans inline
        if row_off is None:
            row_off = torch.empty(n_users + 1, dtype=torch.int32, device=dev)
        L.call("recnn_frame_plan", L.ptr(user_off), L.ptr(batch_users), n_users, frame_size, L.ptr(row_off), None, 0, stream)
    else:
        row_off = None
    out = FrameBatch()
    reward = torch.empty(rows, dtype=torch.float32, device=dev) if reward is None else reward
    done = torch.empty(rows, dtype=torch.float32, device=dev) if done is None else done
    if contiguous:
        state = torch.empty(rows, S, dtype=torch.float32, device=dev)
        nstate = torch.empty(rows, S, dtype=torch.float32, device=dev)
        action = torch.empty(rows, E, dtype=torch.float32, device=dev)
        lds, ldn, lda = S, S, E
    else:
        ld = packed_ld(frame_size, E)
        alloc_rows = (rows + 63) // 64 * 64
        if xs is None:
            xs = torch.empty(alloc_rows, ld, dtype=torch.float32, device=dev)
            xn = torch.empty(alloc_rows, ld, dtype=torch.float32, device=dev)
            xs[:, E + S:].zero_()
            xn[:, E + S:].zero_()
            xn[:, :E].zero_()          # next-action slot, written by the target actor during update()
        state, nstate, action = xs[:rows, E:E + S], xn[:rows, E:E + S], xs[:rows, :E]
        lds = ldn = lda = ld
        out.packed = (xs, xn, reward, done, rows, ld)
    if rows > 0:
        L.call("recnn_frame_gather", L.ptr(items), L.ptr(ratings), L.ptr(user_off), L.ptr(batch_users), L.ptr(row_off),
               n_users, rows, frame_size, E, L.ptr(table), L.ptr(state), lds, L.ptr(nstate), ldn, L.ptr(action), lda,
               L.ptr(reward), L.ptr(done), None, 0, stream)
    out.update(state=state, action=action, reward=reward[:rows], next_state=nstate, done=done[:rows])
    return out


def batch_tensor_embeddings(batch, item_embeddings_tensor, frame_size, *args, **kwargs):
    """Embed Batch: continuous state, continuous action (utils.py:51-81) for an already windowed batch
    {"items": int[B, F+1], "ratings": float[B, F+1], "sizes": int[U], "users": ...}.

    Runs the HIP gather with every window treated as a one-row history; `done` follows utils.py:70-71.
    (FrameEnv does not go through here by default: it gathers straight from the CSR store.)
    """
    items_t, ratings_t, sizes_t, users_t = get_irsu(batch)
    table = item_embeddings_tensor
    _require_cuda(table, "batch_tensor_embeddings(item_embeddings_tensor)")
    dev = table.device
    b = ratings_t.shape[0]
    f1 = frame_size + 1
    items = items_t.to(dev).reshape(-1).to(torch.int32).contiguous()
    ratings = ratings_t.to(dev).reshape(-1).float().contiguous()
    off = torch.arange(b + 1, dtype=torch.int64, device=dev) * f1
    users = torch.arange(b, dtype=torch.int32, device=dev)
    out = gather_frames(items, ratings, off, users, b, b, frame_size, table.float().contiguous(), contiguous=True)
    done = torch.zeros(b, device=dev)
    if b:
        idx = torch.cumsum(torch.as_tensor(sizes_t).to(dev) - frame_size, dim=0) - 1
        idx = idx[(idx >= 0) & (idx < b)]        # (a batch cut to rows_per_batch rows ends inside a user: the later users' last windows are not in it)
        done[idx.long()] = 1
    res = dict(out)
    res["done"] = done
    res["meta"] = {"users": users_t, "sizes": sizes_t}
    return res


def batch_contstate_discaction(batch, item_embeddings_tensor, frame_size, num_items, *args, **kwargs):
    """Embed Batch: continuous state, discrete action (utils.py:84-120): as `batch_tensor_embeddings`, but the action is
    the one-hot row of the item id at the end of the window (float[B, num_items]) -- what REINFORCE's critic and the
    behaviour policy consume.  State / next_state / reward come from the HIP gather, the one-hot rows from
    `recnn_onehot_rows`.  Use with a dataset truncated to `num_items` (`truncate_dataset`): ids must be < num_items."""
    res = batch_tensor_embeddings(batch, item_embeddings_tensor, frame_size)
    from ..nn.functional import onehot_rows
    dev = item_embeddings_tensor.device
    last = torch.as_tensor(batch["items"])[:, -1].to(dev)
    res["action"] = onehot_rows(last, int(num_items))
    return res


def prepare_batch_static_size(batch, item_embeddings_tensor, frame_size=10, embed_batch=batch_tensor_embeddings):
    """The reference's DataLoader collate_fn (utils.py:161-187): list of per-user dicts
    {"items", "rates", "sizes", "users"} -> SARS' batch."""
    sizes = [int(b["sizes"]) for b in batch]
    users_t = torch.tensor([b["users"] for b in batch])
    sizes_t = torch.tensor(sizes)
    if embed_batch is batch_tensor_embeddings:
        table = item_embeddings_tensor
        _require_cuda(table, "prepare_batch_static_size(item_embeddings_tensor)")
        dev = table.device
        items = torch.from_numpy(np.concatenate([np.asarray(b["items"]) for b in batch]).astype(np.int32)).to(dev)
        ratings = torch.from_numpy(np.concatenate([np.asarray(b["rates"]) for b in batch]).astype(np.float32)).to(dev)
        off = np.zeros(len(batch) + 1, dtype=np.int64)
        off[1:] = np.cumsum(sizes)
        rows = int(sum(max(s - frame_size, 0) for s in sizes))
        out = gather_frames(items, ratings, torch.from_numpy(off).to(dev), torch.arange(len(batch), dtype=torch.int32, device=dev),
                            rows, rows, frame_size, table.float().contiguous(), contiguous=True)
        res = dict(out)
        res["meta"] = {"users": users_t, "sizes": sizes_t}
        return res
    # user-supplied embed function: hand it the windowed index batch, as the reference does
    item_t = np.concatenate([rolling_window(np.asarray(b["items"]), frame_size + 1) for b in batch], 0)
    ratings_t = np.concatenate([rolling_window(np.asarray(b["rates"]), frame_size + 1) for b in batch], 0)
    win = {"items": torch.tensor(item_t), "users": users_t, "ratings": torch.tensor(ratings_t).float(), "sizes": sizes_t}
    return embed_batch(batch=win, item_embeddings_tensor=item_embeddings_tensor, frame_size=frame_size)


def make_items_tensor(items_embeddings_key_dict):
    """{item key: tensor[E]} -> (float[N, E] table in sorted-key order, key_to_id, id_to_key)  (utils.py:203-214)."""
    keys = sorted(items_embeddings_key_dict.keys())
    key_to_id = {k: i for i, k in enumerate(keys)}
    id_to_key = {i: k for i, k in enumerate(keys)}
    table = torch.stack([torch.as_tensor(items_embeddings_key_dict[k]) for k in keys])
    return table, key_to_id, id_to_key


def sort_users_itemwise(user_dict, users):
    """User ids ordered by history length, longest first (utils.py:150-156)."""
    return (pd.get().Series({u: user_dict[u]["items"].shape[0] for u in users}).sort_values(ascending=False).index)


def get_base_batch(batch, device=torch.device("cuda"), done=True):
    """[state, action, reward[B,1], next_state, done[B,1]] on `device` (utils.py:265-276).
    Batches made by FrameEnv already live on the GPU, so `.to(device)` moves nothing."""
    b = [batch["state"], batch["action"], batch["reward"].unsqueeze(1), batch["next_state"]]
    if done:
        b.append(batch["done"].unsqueeze(1))
    return [i.to(device) for i in b]
