"""Module-level MLP forward/backward on the HIP GEMM kernels (autograd.Function).

This is the generic path behind `Actor.__call__` / `Critic.__call__` (inference in notebooks, user-defined
losses).  The training step proper does not come through here: `ddpg_update` / `td3_update` run the fused
engine (recnn_amd/nn/fused.py).  fp32 (exact-fp32 MFMA) only; operands are zero-padded to the kernels'
16-byte / 64-element contraction granularity with torch copies (plumbing), bias-gradient column sums come
from the dX kernel's epilogue.
Replaces recnn/nn/models.py:66-73 and :207-213 (addmm, relu, dropout) and their autograd backward.
"""
import ctypes as C
import itertools

import torch

from .. import _lib as L

_call_counter = itertools.count(1)


def _r64(x):
    return (x + 63) // 64 * 64


def _pad(t, rows, cols):
    t = t.detach()
    if t.shape == (rows, cols) and t.is_contiguous() and t.dtype == torch.float32:
        return t
    out = torch.zeros(rows, cols, dtype=torch.float32, device=t.device)
    out[: t.shape[0], : t.shape[1]] = t
    return out


def _args(M, N):
    a = L.GemmArgs()
    a.dtype = L.F32
    a.M, a.N = M, N
    a.dx_scale = 1.0
    a.dw_splits = 1
    return a


def _fwd(x, K, w, bias, out, ldc, N, relu, mask):
    a = _args(x.shape[0], N)
    a.A[0], a.B[0], a.lda[0], a.ldb[0], a.K[0] = x.data_ptr(), w.data_ptr(), x.stride(0), w.stride(0), K
    a.C, a.ldc, a.c_f32 = out.data_ptr(), ldc, 1
    a.bias, a.relu = bias.data_ptr(), int(relu)
    if mask is not None:
        a.mask_mode, a.mask, a.ld_mask = L.MASK_EXTERNAL, mask.data_ptr(), mask.stride(0)
    L.call("recnn_gemm_fwd", C.byref(a), L.current_stream())


def _dx(dz, Kc, w, N, out, yref, scale, colsum):
    a = _args(dz.shape[0], N)
    a.A[0], a.B[0], a.lda[0], a.ldb[0], a.K[0] = dz.data_ptr(), w.data_ptr(), dz.stride(0), w.stride(0), Kc
    a.C, a.ldc, a.c_f32 = out.data_ptr(), out.stride(0), 1
    if yref is not None:
        a.yref, a.ldy, a.dx_scale = yref.data_ptr(), yref.stride(0), scale
    if colsum is not None:
        a.colsum = colsum.data_ptr()
    L.call("recnn_gemm_dx", C.byref(a), L.current_stream())


def _dw(dz, M, x, N, out):
    a = _args(M, N)
    a.A[0], a.B[0], a.lda[0], a.ldb[0], a.K[0] = dz.data_ptr(), x.data_ptr(), dz.stride(0), x.stride(0), dz.shape[0]
    a.C, a.ldc = out.data_ptr(), N
    a.dw_splits, a.dw_slab_stride, a.dw_valid_cols, a.dw_col_rot = 1, M * N, N, 0
    L.call("recnn_gemm_dw", C.byref(a), L.current_stream())


class MLPFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, w3, b3, train, seed):
        if not x.is_cuda:
            raise L.RecnnHipError("recnn_amd networks run on the GPU only (no CPU fallback): move the module and its "
                                  "inputs to 'cuda'")
        B, K = x.shape
        H, O = w1.shape[0], w3.shape[0]
        Kp, Hp, Op = _r64(K), _r64(H), _r64(O)
        dev = x.device
        xp = _pad(x, B, Kp)
        w1p, w2p, w3p = _pad(w1, Hp, Kp), _pad(w2, Hp, Hp), _pad(w3, Op, Hp)
        h1 = torch.zeros(B, Hp, device=dev)
        h2 = torch.zeros(B, Hp, device=dev)
        out = torch.empty(B, O, device=dev)
        m1 = m2 = None
        if train:
            key = next(_call_counter)
            m1 = torch.empty(B, H, dtype=torch.uint8, device=dev)
            m2 = torch.empty(B, H, dtype=torch.uint8, device=dev)
            s = L.current_stream()
            L.call("recnn_hash_mask_dump", seed & 0xFFFFFFFF, key & 0x7FFFFFFF, 0, B, H, L.ptr(m1), s)
            L.call("recnn_hash_mask_dump", seed & 0xFFFFFFFF, key & 0x7FFFFFFF, 1, B, H, L.ptr(m2), s)
        b1c, b2c, b3c = b1.detach().float().contiguous(), b2.detach().float().contiguous(), b3.detach().float().contiguous()
        _fwd(xp, Kp, w1p, b1c, h1, Hp, H, True, m1)
        _fwd(h1, Hp, w2p, b2c, h2, Hp, H, True, m2)
        _fwd(h2, Hp, w3p, b3c, out, O, O, False, None)
        ctx.save_for_backward(xp, h1, h2, w1p, w2p, w3p)
        ctx.dims = (B, K, H, O, Kp, Hp, Op)
        ctx.train = bool(train)
        return out

    @staticmethod
    def backward(ctx, dout):
        xp, h1, h2, w1p, w2p, w3p = ctx.saved_tensors
        B, K, H, O, Kp, Hp, Op = ctx.dims
        dev = dout.device
        scale = 2.0 if ctx.train else 1.0
        tiles = (B + 31) // 32
        doutp = _pad(dout, B, Op)
        gw3 = torch.empty(O, H, device=dev)
        _dw(doutp, O, h2, H, gw3)
        gb3 = dout.sum(0)
        dz2 = torch.zeros(B, Hp, device=dev)
        cs2 = torch.empty(tiles, H, device=dev)
        _dx(doutp, Op, w3p, H, dz2, h2, scale, cs2)
        gw2 = torch.empty(H, H, device=dev)
        _dw(dz2, H, h1, H, gw2)
        dz1 = torch.zeros(B, Hp, device=dev)
        cs1 = torch.empty(tiles, H, device=dev)
        _dx(dz2, Hp, w2p, H, dz1, h1, scale, cs1)
        gw1 = torch.empty(H, K, device=dev)
        _dw(dz1, H, xp, K, gw1)
        gx = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty(B, K, device=dev)
            _dx(dz1, Hp, w1p, K, gx, None, 1.0, None)
        return gx, gw1, cs1.sum(0), gw2, cs2.sum(0), gw3, gb3, None, None


def mlp(x, module, train: bool):
    seed = torch.initial_seed()
    return MLPFunction.apply(x.float(), module.linear1.weight, module.linear1.bias, module.linear2.weight,
                             module.linear2.bias, module.linear3.weight, module.linear3.bias, train, seed)
