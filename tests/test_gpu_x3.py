"""GPU tests of the split-bf16 compute type (RECNN_BF16X3, recnn_amd/csrc/x3.h): every value is hi = bf16(x) + lo = bf16(x - hi),
every product hi*hi + hi*lo + lo*hi on the bf16 matrix cores with fp32 accumulation.

  * the fp32 <-> split-row conversions against their torch restatement, bit for bit;
  * the three GEMM forms (forward: LDS-DMA kernel with the x3 fragment pairing; dX / dW: transpose-read kernels of x3.hip)
    against a float64 product of the SAME split operands (what the kernels are asked to compute: <= 2e-6) and against the
    float64 product of the unsplit fp32 operands (the format's own error: <= 3e-5, bf16 sits at 1e-2);
  * the engine's step in this type against the CPU oracle at the fp32 criteria (tests/test_gpu_engine.py parametrises its
    DDPG / TD3 oracle tests with "bf16x3"; tests/test_gpu_bench_shape.py the 200-step loss curve).
"""
import ctypes as C

import pytest
import torch

from tests.helpers import fro_err, rel_err, x3_cols, x3_pack_ref, x3_unpack_ref, x3_value

pytestmark = pytest.mark.gpu


def _lib():
    from recnn_amd import _lib as L
    L.load()
    return L


def _args(L, M, N):
    a = L.GemmArgs()
    C.memset(C.byref(a), 0, C.sizeof(a))
    a.dtype, a.M, a.N = L.BF16X3, M, N
    a.dx_scale = 1.0
    a.dw_splits = 1
    return a


@pytest.mark.parametrize("R,Cc", [(7, 5), (64, 1418), (33, 256), (5, 32)])
def test_pack_unpack_match_the_torch_restatement(cuda, R, Cc):
    L = _lib()
    g = torch.Generator().manual_seed(R * 1000 + Cc)
    x = torch.randn(R, Cc + 4, generator=g) * torch.logspace(-6, 3, Cc + 4)     # wide exponent range
    xd = x.to(cuda)
    ldp = 2 * ((Cc + 31) // 32 * 32) + 64
    out = torch.zeros(R, ldp, dtype=torch.bfloat16, device=cuda)
    L.call("recnn_x3_pack", L.ptr(xd), Cc + 4, R, Cc, L.ptr(out), ldp, L.current_stream())
    back = torch.full((R, Cc), float("nan"), device=cuda)
    L.call("recnn_x3_unpack", L.ptr(out), ldp, R, Cc, L.ptr(back), Cc, L.current_stream())
    torch.cuda.synchronize()
    ref = x3_pack_ref(x[:, :Cc], ldp)
    assert torch.equal(out.cpu().view(torch.int16), ref.view(torch.int16))
    assert torch.equal(back.cpu(), x3_unpack_ref(ref, Cc))
    # the format itself: 16-17 significant bits
    assert float(((back.cpu() - x[:, :Cc]).abs() / x[:, :Cc].abs().clamp_min(1e-30)).max()) < 2.0 ** -16


@pytest.mark.parametrize("M,N,K", [(2048, 256, 1536), (100, 128, 256), (33, 16, 128), (4096, 256, 256)])
def test_gemm_fwd(cuda, M, N, K):
    """C = dropout(relu(X W^T + b)) with split operands and a split output."""
    L = _lib()
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) * 0.05
    b = torch.randn(N, generator=g)
    mask = (torch.rand(M, N, generator=g) < 0.5).to(torch.uint8)
    X, W = x3_pack_ref(x).to(cuda), x3_pack_ref(w).to(cuda)
    ldc = 2 * ((N + 31) // 32 * 32)
    out = torch.zeros(M, ldc, dtype=torch.bfloat16, device=cuda)
    out32 = torch.zeros(M, N, device=cuda)
    bd, md = b.to(cuda), mask.to(cuda)
    for c_f32 in (0, 1):
        a = _args(L, M, N)
        a.A[0], a.B[0], a.lda[0], a.ldb[0], a.K[0] = X.data_ptr(), W.data_ptr(), 2 * K, 2 * K, 2 * K
        a.C, a.ldc, a.c_f32 = (out32.data_ptr(), N, 1) if c_f32 else (out.data_ptr(), ldc, 0)
        a.bias, a.relu, a.mask_mode, a.mask, a.ld_mask = bd.data_ptr(), 1, L.MASK_EXTERNAL, md.data_ptr(), N
        L.call("recnn_gemm_fwd", C.byref(a), L.current_stream())
    torch.cuda.synchronize()
    xs, ws = x3_value(x).double(), x3_value(w).double()
    ref_split = torch.relu(xs @ ws.t() + b.double()) * mask.double() * 2.0
    ref_full = torch.relu(x.double() @ w.double().t() + b.double()) * mask.double() * 2.0
    assert rel_err(out32, ref_split) < 5e-6       # fp32 accumulation over K (+ the dropped lo*lo products, <= 2^-16 each)
    assert rel_err(out32, ref_full) < 3e-5
    got = x3_unpack_ref(out.cpu(), N)
    assert rel_err(got, ref_full) < 3e-5
    assert torch.equal(got, x3_value(out32.cpu()))                  # the split store of the same accumulators


@pytest.mark.parametrize("M,N,K", [(8192, 256, 1536), (2048, 256, 1536), (333, 256, 256), (20480, 256, 128)])
def test_gemm_fwd_kernel_variants_are_bit_identical(cuda, M, N, K):
    """Every split-bf16 forward GEMM kernel -- round 4's all-waves kernel (x3_fwd = 0), the wave-specialised kernel (2, the default: loader
    waves + consumer waves) with its three tile / ring choices, only-big-launches (11), and the 128 x 128 / 128 x 256 tiles on 128-byte stage
    rows (7, 8) -- computes the same bits: same LDS image, same k order, same epilogue (hash dropout, bias, relu, split store, the policy
    loss's partial dots up to their grouping).  The variant is picked through the private debug hook (csrc/recnn_hip_debug.h)."""
    L = _lib()
    g = torch.Generator().manual_seed(M + K)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) * 0.05
    b = torch.randn(N, generator=g)
    X, W, bd = x3_pack_ref(x).to(cuda), x3_pack_ref(w).to(cuda), b.to(cuda)
    ldc = 2 * ((N + 31) // 32 * 32)
    outs = {}
    try:
        for var in (0, 2, 11, 7, 8):
            L.load().recnn_debug_x3_fwd(var)
            out = torch.zeros(M, ldc, dtype=torch.bfloat16, device=cuda)
            a = _args(L, M, N)
            a.A[0], a.B[0], a.lda[0], a.ldb[0], a.K[0] = X.data_ptr(), W.data_ptr(), 2 * K, 2 * K, 2 * K
            a.C, a.ldc, a.c_f32 = out.data_ptr(), ldc, 0
            a.bias, a.relu, a.mask_mode, a.seed, a.stream_id = bd.data_ptr(), 1, L.MASK_HASH, 77, 3
            L.call("recnn_gemm_fwd", C.byref(a), L.current_stream())
            torch.cuda.synchronize()
            outs[var] = out.view(torch.int16).clone()
    finally:
        L.load().recnn_debug_x3_fwd(-1)
    assert bool((outs[0] != 0).any())
    for var, o in outs.items():
        assert torch.equal(o, outs[0]), var


@pytest.mark.parametrize("M,N,K", [(4096, 256, 256), (8192, 256, 256), (2048, 128, 128)])
def test_gemm_fwd_lean_full_tile_epilogue_equals_the_general_one(cuda, M, N, K):
    """Full tiles of split output leave the wave-specialised kernel through `epilogue_x3_full_tile` (gemm.hip, round 5: the problem's
    fields in registers, no ragged / dot-product / backward-gate code); probe bit 10 (csrc/recnn_hip_debug.h) keeps the general epilogue.
    Element by element the same arithmetic: the same bits under every option the lean form handles (bias, relu, hash / external dropout,
    clamped and unclamped addends), on the 32 x 64 tiles with 3- and 5-stage rings and on 64 x 128 tiles."""
    import itertools
    L = _lib()
    lib = L.load()
    g = torch.Generator().manual_seed(M + K)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) * 0.05
    X, W = x3_pack_ref(x).to(cuda), x3_pack_ref(w).to(cuda)
    bd = torch.randn(N, generator=g).to(cuda)
    ad = (torch.randn(M, N, generator=g) * 3).to(cuda)
    md = (torch.rand(M, N, generator=g) < 0.5).to(torch.uint8).to(cuda)
    ldc = 2 * ((N + 31) // 32 * 32)
    try:
        for use_b, relu, mm, use_add in itertools.product((0, 1), (0, 1), (0, 1, 2), (0, 1, 2)):
            outs = []
            for probe in (0, 1024):
                lib.recnn_debug_x3_ws_probe(probe)
                out = torch.zeros(M, ldc, dtype=torch.bfloat16, device=cuda)
                a = _args(L, M, N)
                a.A[0], a.B[0], a.lda[0], a.ldb[0], a.K[0] = X.data_ptr(), W.data_ptr(), 2 * K, 2 * K, 2 * K
                a.C, a.ldc, a.c_f32, a.relu = out.data_ptr(), ldc, 0, relu
                if use_b:
                    a.bias = bd.data_ptr()
                if mm == 1:
                    a.mask_mode, a.seed, a.stream_id = L.MASK_HASH, 77, 3
                elif mm == 2:
                    a.mask_mode, a.mask, a.ld_mask = L.MASK_EXTERNAL, md.data_ptr(), N
                if use_add:
                    a.addend, a.ld_add, a.add_clip = ad.data_ptr(), N, (1.5 if use_add == 2 else float("inf"))
                L.call("recnn_gemm_fwd", C.byref(a), L.current_stream())
                torch.cuda.synchronize()
                outs.append(out.view(torch.int16).clone())
            assert bool((outs[0] != 0).any())
            assert torch.equal(outs[0], outs[1]), (use_b, relu, mm, use_add)
    finally:
        lib.recnn_debug_x3_ws_probe(0)


def test_gemm_fwd_two_segments(cuda):
    """critic layer 1 on [gen_action | state]: two contraction segments into one accumulator."""
    L = _lib()
    M, N, K0, K1 = 300, 256, 128, 1408
    g = torch.Generator().manual_seed(7)
    x0, x1 = torch.randn(M, K0, generator=g), torch.randn(M, K1, generator=g)
    w = torch.randn(N, K0 + K1, generator=g) * 0.03
    X0, X1, W = x3_pack_ref(x0).to(cuda), x3_pack_ref(x1).to(cuda), x3_pack_ref(w).to(cuda)
    out = torch.zeros(M, N, device=cuda)
    a = _args(L, M, N)
    a.A[0], a.B[0], a.lda[0], a.ldb[0], a.K[0] = X0.data_ptr(), W.data_ptr(), 2 * K0, 2 * (K0 + K1), 2 * K0
    a.A[1], a.B[1], a.lda[1], a.ldb[1], a.K[1] = X1.data_ptr(), W.data_ptr() + 2 * K0 * 2, 2 * K1, 2 * (K0 + K1), 2 * K1
    a.C, a.ldc, a.c_f32 = out.data_ptr(), N, 1
    L.call("recnn_gemm_fwd", C.byref(a), L.current_stream())
    torch.cuda.synchronize()
    ref = torch.cat([x0, x1], 1).double() @ w.double().t()
    assert rel_err(out, ref) < 3e-5


@pytest.mark.parametrize("M,Kc,N", [(2048, 256, 256), (70, 64, 128), (2048, 128, 256), (2048, 256, 128)])
def test_gemm_dx(cuda, M, Kc, N):
    """dX = (dZ W) * scale * [yref > 0] + 32-row column sums; W's split columns run along the TILE dimension."""
    L = _lib()
    g = torch.Generator().manual_seed(M + Kc)
    dz = torch.randn(M, Kc, generator=g)
    w = torch.randn(Kc, N, generator=g) * 0.1
    y = torch.randn(M, N, generator=g)
    DZ, W, Y = x3_pack_ref(dz).to(cuda), x3_pack_ref(w).to(cuda), x3_pack_ref(y).to(cuda)
    ldc = 2 * ((N + 31) // 32 * 32)
    out = torch.zeros(M, ldc, dtype=torch.bfloat16, device=cuda)
    tiles_m = (M + 31) // 32
    colsum = torch.zeros(tiles_m, N, device=cuda)
    a = _args(L, M, N)
    a.A[0], a.B[0], a.lda[0], a.ldb[0], a.K[0] = DZ.data_ptr(), W.data_ptr(), DZ.shape[1], W.shape[1], 2 * Kc
    a.C, a.ldc, a.c_f32 = out.data_ptr(), ldc, 0
    a.yref, a.ldy, a.dx_scale, a.colsum = Y.data_ptr(), Y.shape[1], 2.0, colsum.data_ptr()
    L.call("recnn_gemm_dx", C.byref(a), L.current_stream())
    torch.cuda.synchronize()
    ref = (dz.double() @ w.double()) * 2.0 * (y.double() > 0)
    got = x3_unpack_ref(out.cpu(), N)
    assert rel_err(got, ref) < 3e-5
    ref_slab = ref.view(-1, 32, N).sum(1) if M % 32 == 0 else None
    if ref_slab is not None:
        assert rel_err(colsum, ref_slab) < 3e-5
    assert rel_err(colsum.sum(0), ref.sum(0)) < 3e-5


@pytest.mark.parametrize("rows,M,N,splits", [(2048, 256, 1418, 4), (2048, 256, 256, 16), (333, 128, 256, 2), (50, 32, 27, 1)])
def test_gemm_dw(cuda, rows, M, N, splits):
    """dW slabs = dZ^T X with both operands' split columns along the tile dimensions."""
    L = _lib()
    g = torch.Generator().manual_seed(rows + N)
    dz = torch.randn(rows, M, generator=g) * 1e-3
    x = torch.randn(rows, N, generator=g)
    ldz = 2 * ((M + 63) // 64 * 64)
    ldx = 2 * ((N + 63) // 64 * 64) + 128
    DZ, X = x3_pack_ref(dz, ldz).to(cuda), x3_pack_ref(x, ldx).to(cuda)
    rot = 5 if N > 64 else 0
    slabs = torch.full((splits, M, N), float("nan"), device=cuda)
    a = _args(L, M, N)
    a.A[0], a.B[0], a.lda[0], a.ldb[0], a.K[0] = DZ.data_ptr(), X.data_ptr(), ldz, ldx, rows
    a.C, a.ldc = slabs.data_ptr(), N
    a.dw_splits, a.dw_slab_stride, a.dw_valid_cols, a.dw_col_rot = splits, M * N, N, rot
    L.call("recnn_gemm_dw", C.byref(a), L.current_stream())
    torch.cuda.synchronize()
    got = slabs.sum(0).cpu()
    assert not torch.isnan(got).any()
    ref = torch.roll(dz.double().t() @ x.double(), rot, dims=1)
    assert rel_err(got, ref) < 3e-5
    ref_split = torch.roll(x3_value(dz).double().t() @ x3_value(x).double(), rot, dims=1)
    assert rel_err(got, ref_split) < 6e-6


# ---------------------------------------------------------------- the row-panel launch of layers 2 + 3 (csrc/x3tail.hip)
def _tail_run(L, algo, B, tail, mask_mode, steps=3):
    from tests.test_gpu_engine import _engine, _init_nets, _rand_batch
    S, A, H = 1290, 128, 256
    td3 = algo == "td3"
    actor, critics = _init_nets(8, S, A, H, 2 if td3 else 1)
    batch = _rand_batch(B, S, A, torch.Generator().manual_seed(31))
    eng = _engine(algo, S, A, H, B, "bf16x3", mask_mode=mask_mode, seed=17)
    eng.set_tuning(x3_tail=tail)
    nets = [(L.NET_POLICY, actor), (L.NET_TARGET_POLICY, actor), (L.NET_VALUE1, critics[0]), (L.NET_TARGET_VALUE1, critics[0])]
    if td3:
        nets += [(L.NET_VALUE2, critics[1]), (L.NET_TARGET_VALUE2, critics[1])]
    for ni, p in nets:
        eng.load_params(ni, p)
    eng.set_hyper(policy_opt=dict(lr=1e-4, weight_decay=1e-2), value_opt=dict(lr=1e-4, weight_decay=1e-2), policy_every=2)
    eng.set_counters()
    eng.pack_batch(batch["state"], batch["action"], batch["reward"], batch["next_state"], batch["done"])
    if mask_mode == "external":
        gm = torch.Generator().manual_seed(5)
        masks = [(torch.rand(B, H, generator=gm) < 0.5).to(torch.uint8) for _ in range(8 if td3 else 6)]
        eng.set_external(masks=masks, noise=(torch.randn(B, A, generator=gm) * 0.3) if td3 else None)
    out = []
    names = ("next_action", "gen_action", "expected", "target_q", "q1", "delta1", "critic1_h2", "actor_h2") + (("q2",) if td3 else ())
    for t in range(steps):
        eng.step(B, True, t)
        torch.cuda.synchronize()
        out.append(dict(loss=eng.losses(), bufs={n: eng.buffer(n, B).float().clone() for n in names},
                        p={ni: eng.params[ni].clone() for ni, _ in nets}))
    return out


@pytest.mark.parametrize("algo,B,mask_mode", [("ddpg", 2048, "hash"), ("ddpg", 333, "hash"), ("td3", 1024, "external"), ("ddpg", 31, "none")])
def test_row_panel_tail_equals_per_layer_gemms(cuda, algo, B, mask_mode):
    """x3_tail = 1 (layers 2 + 3 as row-panel launches, h2 kept on chip) against x3_tail = 0 (one grouped GEMM launch per layer):
    the same three-product contraction of the same split operands in a different k order, so every buffer of a step agrees to
    fp32 summation-order distance -- and the dropout pattern, which is a function of (row, column, stream, step) only, exactly."""
    L = _lib()
    ref = _tail_run(L, algo, B, 0, mask_mode)
    new = _tail_run(L, algo, B, 1, mask_mode)
    for t, (a, b) in enumerate(zip(ref, new)):
        for n in a["bufs"]:
            x, y = a["bufs"][n], b["bufs"][n]
            assert rel_err(y, x) < 2e-5, (t, n, rel_err(y, x))
            if n.endswith("_h2") and mask_mode != "none":
                assert torch.equal(x == 0, y == 0) or float(((x == 0) != (y == 0)).float().mean()) < 1e-4, (t, n)
        for k in a["loss"]:
            assert abs(a["loss"][k] - b["loss"][k]) <= 2e-5 * max(1.0, abs(a["loss"][k])), (t, k, a["loss"], b["loss"])
        for ni in a["p"]:
            assert fro_err(b["p"][ni], a["p"][ni]) < 2e-4, (t, ni)     # Adam at the sign-sensitive elements
