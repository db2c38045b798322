"""GPU tests of the DENSE sampler: fixed-row batches that cover the data like the reference's whole-user batches.

  * recnn_frame_plan_dense == its numpy restatement, row for row (carry-in skip, users without windows, the repeated tail slots);
  * the batches the engine's sampler builds from that plan == oracle.frame_batch (the reference's collate, restated) on the same row
    range, bit for bit;
  * `Algo.run(n)` across epoch boundaries == the reference-shaped loop `update(batch); step()` on the materialised dense batches, all
    four networks bit for bit; one epoch's batches hold every (user, window) of the train users exactly once (+ the carried leftover).
"""
import numpy as np
import pytest
import torch

from oracle import recnn_oracle as O
from tests.test_sampler_dense_cpu import FRAME, plan_host

pytestmark = pytest.mark.gpu


def _store(n_users, n_items, seed, lo=8, hi=60):
    rng = np.random.default_rng(seed)
    lens = rng.integers(lo, hi, size=n_users).astype(np.int64)
    off = np.zeros(n_users + 1, dtype=np.int64)
    off[1:] = np.cumsum(lens)
    items = rng.integers(0, n_items, size=int(off[-1]), dtype=np.int32)
    ratings = (2.0 * (rng.integers(1, 11, size=int(off[-1])) * 0.5 - 2.5)).astype(np.float32)
    table = torch.randn(n_items, 128, generator=torch.Generator().manual_seed(seed))
    return items, ratings, off, lens, table


def test_dense_plan_kernel_equals_the_host_restatement(cuda):
    from recnn_amd import _lib as L
    items, ratings, off, lens, table = _store(300, 50, 1)
    lens[5] = FRAME            # a user without a single window
    off = np.concatenate([[0], np.cumsum(lens)])
    rng = np.random.default_rng(2)
    seq = rng.permutation(300).astype(np.int32)
    rows, skip0 = 96, 7
    total = int(np.maximum(lens[seq] - FRAME, 0).sum()) - skip0
    n_e = total // rows
    n_slots = n_e + 2
    off_d, seq_d = torch.from_numpy(off).to(cuda), torch.from_numpy(seq).to(cuda)
    row_off = torch.zeros(len(seq) + 1, dtype=torch.int32, device=cuda)
    plan = torch.full((n_slots * rows,), -7, dtype=torch.int64, device=cuda)
    L.call("recnn_frame_plan_dense", L.ptr(off_d), L.ptr(seq_d), len(seq), skip0, FRAME, rows, L.ptr(row_off), n_slots * rows, L.ptr(plan),
           L.current_stream())
    torch.cuda.synchronize()
    got = plan.cpu().numpy()
    want = plan_host(seq, skip0, off, lens, FRAME, rows, n_e)
    enc = np.array([((off[u] + t) << 1) | d for u, t, d in want], dtype=np.int64)
    assert np.array_equal(got[:n_e * rows], enc)
    assert np.array_equal(got[n_e * rows:], enc[:2 * rows])            # slots past the epoch repeat its first batches
    assert int(row_off[-1].item()) == total


@pytest.mark.parametrize("dtype", ["bf16", "fp32"])
def test_run_over_dense_epochs_equals_the_loop_and_covers_every_window(cuda, dtype):
    import recnn_amd
    from recnn_amd.nn import fused
    n_users, rows = 90, 256
    items, ratings, off, lens, table = _store(n_users, 400, 3, lo=12, hi=70)
    env = recnn_amd.data.env.FrameEnv.from_store(table, items, ratings, off, frame_size=FRAME, batch_size=25, device=cuda, test_fraction=0.0)
    total = int((lens - FRAME).sum())
    n = 3 * (total // rows) + 2                               # three epochs and a bit
    results = {}
    for mode in ("run", "loop"):
        fused.set_defaults(dtype=dtype, mask_mode="hash", seed=77)
        torch.manual_seed(12)
        ddpg = recnn_amd.nn.DDPG(recnn_amd.nn.Actor(1290, 128, 256, 6e-1), recnn_amd.nn.Critic(1290, 128, 256, 54e-2)).to(cuda)
        torch.manual_seed(99)
        ddpg.attach_env(env, rows_per_batch=rows)             # users_per_batch None: dense epochs
        ctx = ddpg._fused_ctx
        assert ctx.sampler["dense"] and ctx.sampler["n_batches"] in (total // rows, total // rows + 1)
        if mode == "run":
            ddpg.run(7)
            ddpg.run(n - 7)
            seen = None
        else:
            seen = {}
            tab = table.numpy()
            for i, batch in enumerate(ddpg.batches(n)):
                rows_i = {k: batch[k] for k in ("state", "action", "reward", "next_state", "done")}   # materialised on demand
                # ---- the same row range through the reference's collate (restated): whole users, then the cut
                sm = ctx.sampler
                ep = max(e for e, p0 in sm["epoch_pos"].items() if p0 <= i)
                seq, skip0, n_e = sm["seqs"][ep]
                idx = i - sm["epoch_pos"][ep]
                wins = np.maximum(lens[seq] - FRAME, 0)
                wins[0] -= skip0
                cum = np.cumsum(wins)
                i0 = int(np.searchsorted(cum, idx * rows, side="right"))
                i1 = int(np.searchsorted(cum, idx * rows + rows - 1, side="right"))
                a = idx * rows - (int(cum[i0 - 1]) if i0 else 0) + (skip0 if i0 == 0 else 0)
                us = seq[i0:i1 + 1]
                ref = O.frame_batch([items[off[u]:off[u + 1]].astype(np.int64) for u in us], [ratings[off[u]:off[u + 1]].astype(np.float64) for u in us],
                                    tab, FRAME)
                for k in ("state", "action", "reward", "next_state", "done"):
                    assert np.array_equal(rows_i[k].float().cpu().numpy().reshape(rows, -1), np.asarray(ref[k][a:a + rows]).reshape(rows, -1)), (i, k)
                # ---- coverage bookkeeping: (user, window) of every row of this batch
                lst = [(int(u), t) for j, u in enumerate(us) for t in range(a if j == 0 else 0, int(lens[u]) - FRAME)][:rows]
                assert len(lst) == rows
                for key in lst:
                    seen.setdefault(key, []).append(i)
                ddpg.update(batch, learn=True)
                ddpg.step()
            ddpg.flush()
        torch.cuda.synchronize()
        assert ddpg._step == n
        results[mode] = ({nm: {k: v.detach().clone() for k, v in ddpg.nets[nm].state_dict().items()}
                          for nm in ("policy_net", "value_net", "target_policy_net", "target_value_net")}, seen, ctx)
    for net, sd in results["run"][0].items():
        for k, v in sd.items():
            assert torch.equal(v, results["loop"][0][net][k]), (net, k)
    # ---- coverage: over the first three epochs every (user, window) appears three times, counting what the third epoch carried over
    seen, ctx = results["loop"][1], results["loop"][2]
    sm = ctx.sampler
    first3 = sm["epoch_pos"][3] if 3 in sm["epoch_pos"] else None
    assert first3 is not None and first3 <= n
    counts = {}
    for key, where in seen.items():
        counts[key] = sum(1 for b in where if b < first3)
    seq3, skip3, _ = sm["seqs"][3]
    carried = {}
    n_new = n_users
    carry_slots = seq3[:len(seq3) - n_new]
    for j, u in enumerate(carry_slots):
        for t in range(skip3 if j == 0 else 0, int(lens[u]) - FRAME):
            carried[(int(u), t)] = 1
    for u in range(n_users):
        for t in range(int(lens[u]) - FRAME):
            assert counts.get((u, t), 0) + carried.get((u, t), 0) == 3, (u, t, counts.get((u, t), 0), carried.get((u, t), 0))
