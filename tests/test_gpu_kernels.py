"""GPU parity tests of the individual HIP kernels, called through the C ABI (ctypes).

gather: bit-exact against the committed reference fixtures and the CPU oracle.
GEMMs : fp32 MFMA path within 1e-5 of a float64 reference, bf16 path within bf16 rounding.
"""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from oracle import recnn_oracle as O
from tests.helpers import csr, make_store, rel_err

pytestmark = pytest.mark.gpu


def _lib():
    from recnn_amd import _lib as L
    L.load()
    return L


def _gather(L, dev, items, ratings, table, users, frame, rows=None, packed=False, rows_per_wg=None, inline_plan=False):
    it, rt, off = csr(items, ratings)
    E = table.shape[1]
    S = frame * E + frame
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    it_d, rt_d, off_d, tab_d = d(it), d(rt), d(off), d(table)
    users_d = torch.tensor(users, dtype=torch.int32, device=dev)
    row_off = torch.zeros(len(users) + 1, dtype=torch.int32, device=dev)
    L.call("recnn_frame_plan", L.ptr(off_d), L.ptr(users_d), len(users), frame, L.ptr(row_off), None, 0, L.current_stream())
    total = int(row_off[-1].item())
    B = total if rows is None else rows
    if packed:
        ld = ((E + ((S + 63) // 64) * 64) + 63) // 64 * 64
        xs = torch.full((B, ld), 7.0, device=dev)
        xn = torch.full((B, ld), 7.0, device=dev)
        state, nstate, action = xs[:, E:E + S], xn[:, E:E + S], xs[:, :E]
        lds = ldn = lda = ld
    else:
        state = torch.full((B, S), 7.0, device=dev)
        nstate = torch.full((B, S), 7.0, device=dev)
        action = torch.full((B, E), 7.0, device=dev)
        lds = ldn = S
        lda = E
    reward = torch.full((B,), 7.0, device=dev)
    done = torch.full((B,), 7.0, device=dev)
    L.call("recnn_frame_gather", L.ptr(it_d), L.ptr(rt_d), L.ptr(off_d), L.ptr(users_d), None if inline_plan else L.ptr(row_off), len(users), B,
           frame, E, L.ptr(tab_d), L.ptr(state), lds, L.ptr(nstate), ldn, L.ptr(action), lda, L.ptr(reward), L.ptr(done),
           None, 0, L.current_stream())
    torch.cuda.synchronize()
    return dict(state=state.cpu().numpy(), next_state=nstate.cpu().numpy(), action=action.cpu().numpy(),
                reward=reward.cpu().numpy(), done=done.cpu().numpy(), total=total)


@pytest.mark.parametrize("name", ["tiny", "f10e128"])
@pytest.mark.parametrize("packed", [False, True])
def test_gather_matches_reference_fixture(cuda, golden_dir, name, packed):
    L = _lib()
    g = np.load(os.path.join(golden_dir, f"gather_{name}.npz"))
    lens = g["lengths"]
    frame = int(g["frame"])
    offs = np.concatenate([[0], np.cumsum(lens)])
    items = [g["items_flat"][offs[i]:offs[i + 1]] for i in range(len(lens))]
    ratings = [g["ratings_flat"][offs[i]:offs[i + 1]] for i in range(len(lens))]
    out = _gather(L, cuda, items, ratings, g["table"], list(range(len(lens))), frame, packed=packed)
    for k in ("state", "next_state", "action", "reward", "done"):
        assert np.array_equal(out[k], g[k]), k          # bit-exact (pure copy / integer work)


def test_gather_bitexact_vs_oracle_b2048(cuda, rows_per_wg=None):
    """BASELINE config shape: F=10, E=128, exactly 2048 rows (last user truncated), shuffled users."""
    L = _lib()
    items, ratings, table = make_store(n_users=60, n_items=5000, emb_dim=128, min_len=11, max_len=120, seed=3)
    rng = np.random.default_rng(0)
    users = rng.permutation(60)[:50].tolist()
    ref = O.frame_batch([items[u] for u in users], [ratings[u] for u in users], table, 10, rows=2048)
    assert ref["state"].shape[0] == 2048
    for packed in (False, True):
        for inline_plan in (False, True):
            out = _gather(L, cuda, items, ratings, table, users, 10, rows=2048, packed=packed, rows_per_wg=rows_per_wg,
                          inline_plan=inline_plan)
            for k in ("state", "next_state", "action", "reward", "done"):
                assert np.array_equal(out[k], ref[k]), (k, packed, inline_plan)


def test_gather_edge_cases(cuda):
    L = _lib()
    # users with exactly F+1 interactions (one row each), a single user, and a one-row request
    items, ratings, table = make_store(n_users=7, n_items=40, emb_dim=16, min_len=6, max_len=6, seed=5)
    for users, rows in (([3], None), (list(range(7)), None), ([6, 0, 2], 1), ([1, 1, 1], None)):
        ref = O.frame_batch([items[u] for u in users], [ratings[u] for u in users], table, 5, rows=rows)
        for inline_plan in (False, True):
            out = _gather(L, cuda, items, ratings, table, users, 5, rows=rows, inline_plan=inline_plan)
            for k in ("state", "next_state", "action", "reward", "done"):
                assert np.array_equal(out[k], ref[k]), (k, users, inline_plan)


# ------------------------------------------------------------------------------------------- GEMMs
def _tc(t, dtype):
    return t.float().contiguous() if dtype == "fp32" else t.to(torch.bfloat16).contiguous()


def _tol(dtype):
    return 2e-5 if dtype == "fp32" else 2e-2


def _args(L, dtype, M, N):
    a = L.GemmArgs()
    a.dtype = L.F32 if dtype == "fp32" else L.BF16
    a.M, a.N = M, N
    a.dx_scale = 1.0
    a.dw_splits = 1
    return a


@pytest.mark.parametrize("upb,rows", [(7, 300), (256, 2048), (1500, 96)])
def test_plan_table_rows_match_the_window_arithmetic(cuda, upb, rows):
    """recnn_frame_plan_rows (one workgroup per batch of an epoch permutation): plan[b][r] = (CSR offset of the window start
    of row r << 1) | done, -1 past the batch's rows -- against the prefix-sum / search arithmetic of
    prepare_batch_static_size in numpy (users with fewer than frame + 1 ratings contribute no rows, repeated users,
    more than 256 users per batch)."""
    L = _lib()
    rng = np.random.default_rng(upb)
    n_store, frame, n_batches = 400, 10, 5
    lens = rng.integers(3, 60, n_store)                       # some histories are shorter than a window
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    perm = rng.integers(0, n_store, n_batches * upb).astype(np.int32)       # with repeats
    plan = torch.full((n_batches * rows,), 123, dtype=torch.int64, device=cuda)
    off_d, perm_d = torch.from_numpy(off).to(cuda), torch.from_numpy(perm).to(cuda)       # (kept alive across the launch)
    L.call("recnn_frame_plan_rows", L.ptr(off_d), L.ptr(perm_d), upb, n_batches, frame, rows, L.ptr(plan), L.current_stream())
    torch.cuda.synchronize()
    got = plan.cpu().numpy().reshape(n_batches, rows)
    for b in range(n_batches):
        want = []
        for u in perm[b * upb:(b + 1) * upb]:
            n = max(int(lens[u]) - frame, 0)
            want += [((int(off[u]) + t) << 1) | (1 if t == n - 1 else 0) for t in range(n)]
        want = (want + [-1] * rows)[:rows]
        want = np.asarray(want, dtype=np.int64)
        bad = np.flatnonzero(got[b] != want)
        assert bad.size == 0, (b, bad[:8], got[b][bad[:8]], want[bad[:8]])


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("M,N,K", [(2048, 256, 1472), (100, 128, 256), (33, 16, 64), (4096, 256, 256)])
def test_gemm_fwd(cuda, dtype, M, N, K):
    L = _lib()
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) * 0.05        # asymmetric operands (transposition would be caught)
    b = torch.randn(N, generator=g)
    mask = (torch.rand(M, N, generator=g) < 0.5).to(torch.uint8)
    X, W = _tc(x, dtype).to(cuda), _tc(w, dtype).to(cuda)
    out = torch.zeros(M, N, device=cuda)
    a = _args(L, dtype, M, N)
    a.A[0], a.B[0], a.lda[0], a.ldb[0], a.K[0] = X.data_ptr(), W.data_ptr(), K, K, K
    a.C, a.ldc, a.c_f32 = out.data_ptr(), N, 1
    bd, md = b.to(cuda), mask.to(cuda)
    a.bias, a.relu, a.mask_mode, a.mask, a.ld_mask = bd.data_ptr(), 1, L.MASK_EXTERNAL, md.data_ptr(), N
    L.call("recnn_gemm_fwd", C.byref(a), L.current_stream())
    torch.cuda.synchronize()
    ref = torch.relu(X.double().cpu() @ W.double().cpu().t() + b.double()) * mask.double() * 2.0
    assert rel_err(out, ref) < _tol(dtype)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("M,N,K,tc_out", [(256, 512, 40960, False), (200, 300, 33024, True), (128, 130, 65536, False)])
def test_gemm_fwd_split_k(cuda, dtype, M, N, K, tc_out):
    """recnn_gemm_args::ws: a catalogue-long contraction with few output tiles is cut into K slices (gemm.hip launch_dma_splitk);
    the second launch sums them in slice order and applies the whole forward epilogue (bias, clamped addend, relu, the backward
    gate, external dropout mask, fp32 or compute-type output).  Against the float64 product, and against the same call without
    scratch (one workgroup per tile): equal up to fp32 summation order.  Deterministic: two calls, the same bits."""
    L = _lib()
    g = torch.Generator().manual_seed(M + N)
    x = torch.randn(M, K, generator=g) * 0.1
    w = torch.randn(N, K, generator=g) * 0.05
    b = torch.randn(N, generator=g)
    add = torch.randn(M, N, generator=g) * 3.0
    y = torch.randn(M, N, generator=g)
    mask = (torch.rand(M, N, generator=g) < 0.5).to(torch.uint8)
    X, W = _tc(x, dtype).to(cuda), _tc(w, dtype).to(cuda)
    Y = _tc(y, dtype).to(cuda)
    bd, md, ad = b.to(cuda), mask.to(cuda), add.to(cuda)
    ws = torch.empty(8 * M * N, device=cuda)
    outs = []
    for use_ws in (True, True, False):
        out = torch.zeros(M, N, device=cuda, dtype=(torch.float32 if dtype == "fp32" else torch.bfloat16) if tc_out else torch.float32)
        a = _args(L, dtype, M, N)
        a.A[0], a.B[0], a.lda[0], a.ldb[0], a.K[0] = X.data_ptr(), W.data_ptr(), K, K, K
        a.C, a.ldc, a.c_f32 = out.data_ptr(), N, 0 if tc_out else 1
        a.bias, a.relu, a.mask_mode, a.mask, a.ld_mask = bd.data_ptr(), 1, L.MASK_EXTERNAL, md.data_ptr(), N
        a.addend, a.ld_add, a.add_clip = ad.data_ptr(), N, 1.5
        a.yref, a.ldy, a.dx_scale = Y.data_ptr(), N, 0.5
        if use_ws:
            a.ws, a.ws_bytes = ws.data_ptr(), ws.numel() * 4
        L.call("recnn_gemm_fwd", C.byref(a), L.current_stream())
        torch.cuda.synchronize()
        outs.append(out.float().cpu())
    ref = torch.relu(X.double().cpu() @ W.double().cpu().t() + b.double() + add.double().clamp(-1.5, 1.5))
    ref = torch.where(Y.double().cpu() > 0, ref * 0.5, torch.zeros_like(ref)) * mask.double() * 2.0
    tol = _tol(dtype) if not (tc_out and dtype == "bf16") else 1e-2
    assert rel_err(outs[0], ref) < tol
    assert torch.equal(outs[0], outs[1])
    assert rel_err(outs[0], outs[2]) < (1e-5 if not (tc_out and dtype == "bf16") else 1e-2)


@pytest.mark.parametrize("M,N,K,tc_out", [(256, 9000, 2048, False), (300, 8200, 1344, True), (129, 8192, 128, False)])
def test_gemm_fwd_catalogue_wide_tall_tile_equals_the_square_tile(cuda, M, N, K, tc_out):
    """Catalogue-wide bf16 products of more than 128 rows (REINFORCE: [256, 2048] x [2048, 100k]) run as 256 x 128 tiles on the
    wave-specialised kernel (gemm.hip launch_dma_wide, round 5); `recnn_debug_wide_ws(0)` keeps round 3's 128 x 128 kernel.  Same products
    in the same k order: the two must agree BIT FOR BIT (ragged last row / column tiles included), and with the float64 product."""
    L = _lib()
    lib = L.load()
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) * 0.05
    b = torch.randn(N, generator=g)
    X, W, bd = _tc(x, "bf16").to(cuda), _tc(w, "bf16").to(cuda), b.to(cuda)
    outs = []
    try:
        for tall in (1, 0):
            lib.recnn_debug_wide_ws(tall)
            out = torch.zeros(M, N, device=cuda, dtype=torch.bfloat16 if tc_out else torch.float32)
            a = _args(L, "bf16", M, N)
            a.A[0], a.B[0], a.lda[0], a.ldb[0], a.K[0] = X.data_ptr(), W.data_ptr(), K, K, K
            a.C, a.ldc, a.c_f32 = out.data_ptr(), N, 0 if tc_out else 1
            a.bias, a.relu = bd.data_ptr(), 1
            L.call("recnn_gemm_fwd", C.byref(a), L.current_stream())
            torch.cuda.synchronize()
            outs.append(out.float().cpu())
    finally:
        lib.recnn_debug_wide_ws(1)
    ref = torch.relu(X.double().cpu() @ W.double().cpu().t() + b.double())
    assert rel_err(outs[0], ref) < (1e-2 if tc_out else 2e-2)
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_gemm_fwd_two_segments_f32_inputs_tc_output(cuda, dtype):
    """critic layer 1 on [gen_action | state]: 2 contraction segments, fp32 packed inputs, tc output."""
    L = _lib()
    M, N, K0, K1 = 300, 256, 128, 1344
    g = torch.Generator().manual_seed(7)
    x0, x1 = torch.randn(M, K0, generator=g), torch.randn(M, 1500, generator=g)
    w = torch.randn(N, K0 + K1, generator=g) * 0.03
    X0, X1, W = x0.to(cuda), x1.to(cuda), _tc(w, dtype).to(cuda)
    out = torch.zeros(M, N, device=cuda, dtype=torch.float32 if dtype == "fp32" else torch.bfloat16)
    a = _args(L, dtype, M, N)
    a.A[0], a.B[0], a.lda[0], a.ldb[0], a.K[0], a.a_f32[0] = X0.data_ptr(), W.data_ptr(), K0, K0 + K1, K0, 1
    esz = 4 if dtype == "fp32" else 2
    a.A[1], a.B[1], a.lda[1], a.ldb[1], a.K[1], a.a_f32[1] = X1.data_ptr(), W.data_ptr() + K0 * esz, 1500, K0 + K1, K1, 1
    a.C, a.ldc, a.c_f32 = out.data_ptr(), N, 0
    L.call("recnn_gemm_fwd", C.byref(a), L.current_stream())
    torch.cuda.synchronize()
    xin = torch.cat([x0, x1[:, :K1]], 1)
    if dtype == "bf16":
        xin = xin.to(torch.bfloat16)
    ref = xin.double() @ W.double().cpu().t()
    assert rel_err(out, ref) < (_tol(dtype) if dtype == "fp32" else 3e-2)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("M,Kc,N", [(2048, 256, 256), (70, 64, 128), (2048, 128, 256)])
def test_gemm_dx(cuda, dtype, M, Kc, N):
    L = _lib()
    g = torch.Generator().manual_seed(M + Kc)
    dz = torch.randn(M, Kc, generator=g)
    w = torch.randn(Kc, N, generator=g) * 0.1
    y = torch.randn(M, N, generator=g)
    DZ, W, Y = _tc(dz, dtype).to(cuda), _tc(w, dtype).to(cuda), _tc(y, dtype).to(cuda)
    out = torch.zeros(M, N, device=cuda, dtype=DZ.dtype)
    tiles_m = (M + 31) // 32
    colsum = torch.zeros(tiles_m, N, device=cuda)
    a = _args(L, dtype, M, N)
    a.A[0], a.B[0], a.lda[0], a.ldb[0], a.K[0] = DZ.data_ptr(), W.data_ptr(), Kc, N, Kc
    a.C, a.ldc, a.c_f32 = out.data_ptr(), N, 0
    a.yref, a.ldy, a.dx_scale, a.colsum = Y.data_ptr(), N, 2.0, colsum.data_ptr()
    L.call("recnn_gemm_dx", C.byref(a), L.current_stream())
    torch.cuda.synchronize()
    ref = (DZ.double().cpu() @ W.double().cpu()) * 2.0 * (Y.double().cpu() > 0)
    assert rel_err(out, ref) < _tol(dtype)
    assert rel_err(colsum.sum(0), ref.sum(0)) < (_tol(dtype) if dtype == "fp32" else 5e-2)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("rows,M,N,splits,b32", [(2048, 256, 1418, 4, True), (2048, 256, 256, 16, False),
                                                 (333, 128, 256, 2, False), (50, 16, 27, 1, True)])
def test_gemm_dw(cuda, dtype, rows, M, N, splits, b32):
    L = _lib()
    g = torch.Generator().manual_seed(rows + N)
    ldz = (M + 63) // 64 * 64
    ldx = (N + 63) // 64 * 64 + 64
    dz = torch.zeros(rows, ldz)
    dz[:, :M] = torch.randn(rows, M, generator=g)
    x = torch.zeros(rows, ldx)
    x[:, :N] = torch.randn(rows, N, generator=g)
    DZ = _tc(dz, dtype).to(cuda)
    X = x.to(cuda) if (b32 or dtype == "fp32") else _tc(x, dtype).to(cuda)
    rot = 5 if N > 64 else 0
    slabs = torch.full((splits, M, N), float("nan"), device=cuda)
    a = _args(L, dtype, M, N)
    a.A[0], a.B[0], a.lda[0], a.ldb[0], a.K[0] = DZ.data_ptr(), X.data_ptr(), ldz, ldx, rows
    a.b_f32[0] = 1 if (b32 and dtype == "bf16") else 0
    a.C, a.ldc = slabs.data_ptr(), N
    a.dw_splits, a.dw_slab_stride, a.dw_valid_cols, a.dw_col_rot = splits, M * N, N, rot
    L.call("recnn_gemm_dw", C.byref(a), L.current_stream())
    torch.cuda.synchronize()
    got = slabs.sum(0).cpu()
    assert not torch.isnan(got).any()
    xr = X.double().cpu()[:, :N]
    ref = DZ.double().cpu()[:, :M].t() @ xr
    ref = torch.roll(ref, rot, dims=1)            # stored column = (col + rot) mod N
    assert rel_err(got, ref) < _tol(dtype)


def test_hash_mask_is_fair_and_reproducible(cuda):
    L = _lib()
    m1 = torch.zeros(2048, 256, dtype=torch.uint8, device=cuda)
    m2 = torch.zeros_like(m1)
    m3 = torch.zeros_like(m1)
    L.call("recnn_hash_mask_dump", 123, 5, 2, 2048, 256, L.ptr(m1), L.current_stream())
    L.call("recnn_hash_mask_dump", 123, 5, 2, 2048, 256, L.ptr(m2), L.current_stream())
    L.call("recnn_hash_mask_dump", 123, 6, 2, 2048, 256, L.ptr(m3), L.current_stream())
    torch.cuda.synchronize()
    assert torch.equal(m1, m2)
    p = m1.float().mean().item()
    assert abs(p - 0.5) < 0.01
    assert abs((m1 ^ m3).float().mean().item() - 0.5) < 0.01          # independent across steps
    assert abs(m1.float().mean(0).std().item()) < 0.03                  # no dead/always-on columns
    # the mask the fwd epilogue applies is exactly this dump
    x = torch.ones(2048, 64, device=cuda)
    w = torch.ones(256, 64, device=cuda)
    out = torch.zeros(2048, 256, device=cuda)
    step = torch.tensor([5], dtype=torch.int32, device=cuda)
    a = _args(L, "fp32", 2048, 256)
    a.A[0], a.B[0], a.lda[0], a.ldb[0], a.K[0] = x.data_ptr(), w.data_ptr(), 64, 64, 64
    a.C, a.ldc, a.c_f32 = out.data_ptr(), 256, 1
    a.mask_mode, a.seed, a.stream_id, a.step_ptr = L.MASK_HASH, 123, 2, step.data_ptr()
    L.call("recnn_gemm_fwd", C.byref(a), L.current_stream())
    torch.cuda.synchronize()
    assert torch.equal(out, m1.float() * 128.0)


def test_flat_optimizer_kernels(cuda):
    L = _lib()
    n = 429_313
    g = torch.Generator().manual_seed(1)
    p, gr = torch.randn(n, generator=g), torch.randn(n, generator=g) * 1e-3
    P = {"w1": p.clone()}
    st = O.AdamState(lr=1e-3, weight_decay=1e-2)
    pd, gd = p.to(cuda), gr.to(cuda)
    m, v = torch.zeros(n, device=cuda), torch.zeros(n, device=cuda)
    old_order = O.PARAM_ORDER
    O.PARAM_ORDER = ("w1",)
    try:
        for t in (1, 2, 3):
            L.call("recnn_adam_flat", L.ptr(pd), L.ptr(gd), L.ptr(m), L.ptr(v), n, 1e-3, 0.9, 0.999, 1e-8, 1e-2, t, -0.5,
                   L.current_stream())
            O.adam_step(P, {"w1": gr}, st, grad_scale=-0.5)
    finally:
        O.PARAM_ORDER = old_order
    torch.cuda.synchronize()
    assert rel_err(pd, P["w1"]) < 1e-6
    tgt = torch.randn(n, generator=g)
    td = tgt.to(cuda)
    L.call("recnn_soft_update_flat", L.ptr(td), L.ptr(pd), n, 0.001, L.current_stream())
    assert rel_err(td, tgt * (1 - 0.001) + pd.cpu() * 0.001) < 1e-6
    scratch, out = torch.zeros(1024, device=cuda), torch.zeros(1, device=cuda)
    L.call("recnn_l1_norm_flat", L.ptr(gd), n, L.ptr(scratch), L.ptr(out), L.current_stream())
    assert abs(out.item() / gr.double().abs().sum().item() - 1) < 1e-5


@pytest.mark.parametrize("R,Cc,bf16", [(1000, 130, False), (4099, 64, True), (63, 257, True), (1, 1, False)])
def test_transpose_rows(cuda, R, Cc, bf16):
    """recnn_transpose_rows: dst[c, r] = src[r, c] (exact in fp32, round-to-nearest-even in bf16), padding columns untouched."""
    L = _lib()
    src = torch.randn(R, Cc + 3, device=cuda)[:, :Cc]           # a strided source (row stride cols + 3)
    ldt = (R + 7) // 8 * 8 + 8
    dst = torch.full((Cc, ldt), 5.0, device=cuda, dtype=torch.bfloat16 if bf16 else torch.float32)
    L.call("recnn_transpose_rows", L.ptr(src), src.stride(0), R, Cc, L.ptr(dst), ldt, int(bf16), L.current_stream())
    torch.cuda.synchronize()
    assert torch.equal(dst[:, :R], src.t().to(dst.dtype))
    assert bool((dst[:, R:] == 5).all())
