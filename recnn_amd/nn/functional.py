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

# Dropout-mask keys on the device (captured updates, recnn_amd/nn/graphed.py): while a key scope is open, every train-mode MLP call
# keys its two hash masks with `*dev + add` (add = calls so far in the scope) instead of the host-side call counter, so that a
# captured graph draws fresh masks on every replay; the scope's owner advances *dev by the calls consumed.
_key_scope = {"dev": None, "add": 0}


def open_key_scope(dev_counter):
    _key_scope["dev"], _key_scope["add"] = dev_counter, 0


def close_key_scope() -> int:
    used = _key_scope["add"]
    _key_scope["dev"], _key_scope["add"] = None, 0
    return used


def _dump_masks(seed, B, H, m1, m2, s):
    if _key_scope["dev"] is not None:
        add = _key_scope["add"]
        _key_scope["add"] = add + 1
        for stream_id, m in ((0, m1), (1, m2)):
            L.call("recnn_hash_mask_dump_at", seed & 0xFFFFFFFF, L.ptr(_key_scope["dev"]), add, stream_id, B, H, L.ptr(m), s)
        return
    key = next(_call_counter)
    L.call("recnn_hash_mask_dump", seed & 0xFFFFFFFF, key & 0x7FFFFFFF, 0, B, H, L.ptr(m1), s)
    L.call("recnn_hash_mask_dump", seed & 0xFFFFFFFF, key & 0x7FFFFFFF, 1, B, H, L.ptr(m2), s)


def _r64(x):
    return (x + 63) // 64 * 64


def _pad(t, rows, cols):
    t = t.detach()
    if t.shape == (rows, cols) and t.is_contiguous() and t.dtype == torch.float32:
        return t
    out = torch.zeros(rows, cols, dtype=torch.float32, device=t.device)
    out[: t.shape[0], : t.shape[1]] = t
    return out


_derived = {}
_written = {}     # id(parameter) -> number of writes the HIP engine made behind autograd's back (mark_written)


_capture_cache = {"d": None}


def open_capture_cache():
    _capture_cache["d"] = {}


def close_capture_cache():
    _capture_cache["d"] = None


def mark_written(params):
    """The fused engine (ddpg_update / td3_update / value_update / Algo.run) writes adopted parameters in place from its
    kernels: no torch op runs, so their autograd version counters do not move.  Every caller that lets the engine step a
    network reports its parameters here; `_derived_of` folds the count into its staleness tag, so a cached bf16 shadow /
    padded / transposed copy made before the step is rebuilt at the next module-level forward.  (A user's own
    `p.data.copy_()` is invisible to both counters -- `.data` carries its own version; call
    `fused.notify_params_changed(module)` after such a write.)"""
    for p in params:
        k = id(p)
        _written[k] = _written.get(k, 0) + 1


def _derived_of(w, kind, build):
    """A layout derived from parameter `w` (transposed / padded copy), kept until `w` is written again.  Staleness is
    detected through the autograd version counter -- torch's in-place ops bump it, and so do recnn_amd's own torch-side
    writers (optim.Adam / Ranger, utils.soft_update call torch.autograd.graph.increment_version) -- plus the engine's
    write count (`mark_written`) for the kernels that update adopted parameters without any torch op."""
    import weakref
    if w.is_cuda and torch.cuda.is_current_stream_capturing():
        # a captured update (recnn_amd/nn/graphed.py) must rebuild the layout on every replay: a hit in the long-lived cache would
        # leave the build out of the graph, and the graph would keep reading the copy made before the capture after `w` moved on.
        # Inside ONE capture the same version of `w` is built once (a cache that lives as long as the capture: every replay
        # rebuilds at the same points of the step).
        tag = (w._version, w.data_ptr(), tuple(w.shape), _written.get(id(w), 0))
        scoped = _capture_cache.get("d")
        if scoped is None:
            return build(w.detach())
        hit = scoped.get((id(w), kind))
        if hit is not None and hit[0] == tag:
            return hit[1]
        out = build(w.detach())
        scoped[(id(w), kind)] = (tag, out, w)      # (w kept alive: its id must not be reused while the capture runs)
        return out
    key = (id(w), kind)
    tag = (w._version, w.data_ptr(), tuple(w.shape), _written.get(id(w), 0))
    hit = _derived.get(key)
    if hit is not None and hit[0]() is w and hit[1] == tag:
        return hit[2]
    out = build(w.detach())
    _derived[key] = (weakref.ref(w, lambda _r, k=key: (_derived.pop(k, None), _written.pop(k[0], None))), tag, out)
    return out


def shadow_target(p):
    """(kind, tensor) of a cached row-padded copy of the 2-D parameter `p` (same rows and columns, wider leading dimension, fp32 or
    bf16: the "padded" / "bf16..." layouts above; transposed copies do not qualify) that is CURRENT -- an optimizer kernel that
    writes `p` can rewrite it in the same pass (`recnn_*_flat_shadow`) instead of leaving a full conversion pass to the next
    forward -- or None.  The caller reports the write with `shadow_written`."""
    if p.dim() != 2 or not p.is_cuda or torch.cuda.is_current_stream_capturing():
        return None
    tag = (p._version, p.data_ptr(), tuple(p.shape), _written.get(id(p), 0))
    best = None
    for (pid, kind), (ref, have, out) in _derived.items():
        if pid != id(p) or ref() is not p or have != tag or kind.startswith("transposed"):
            continue
        if out.dim() != 2 or out.stride(1) != 1 or out.shape[0] < p.shape[0] or out.shape[1] < p.shape[1]:
            continue
        if out.dtype not in (torch.float32, torch.bfloat16) or out.data_ptr() == p.data_ptr():
            continue
        if best is None or (out.dtype == torch.bfloat16 and best[1].dtype != torch.bfloat16):
            best = (kind, out)
    return best


def shadow_written(p, kind):
    """`p` was stepped (its version already bumped) and the kernel rewrote the cached layout `kind`: it is current again."""
    key = (id(p), kind)
    hit = _derived.get(key)
    if hit is not None and hit[0]() is p:
        _derived[key] = (hit[0], (p._version, p.data_ptr(), tuple(p.shape), _written.get(id(p), 0)), hit[2])


_catalogue_dtype = "fp32"


def set_catalogue_dtype(dtype: str):
    """Compute type of the catalogue-sized GEMMs of the REINFORCE path ([rows, hidden] x [hidden, n_items] of the policy
    head and its backward, [rows, state + n_items] x [., hidden] of the critic over action distributions, the logits GEMM of the
    learned behaviour policy `Beta`):
    'fp32' (default; exact-fp32 MFMA, the parity mode) or 'bf16' (bf16 MFMA with fp32 accumulation over bf16 copies of the
    weights kept per weight version; softmax, log-prob, optimizer and all small layers stay fp32)."""
    global _catalogue_dtype
    if dtype not in ("fp32", "bf16"):
        raise ValueError(dtype)
    _catalogue_dtype = dtype


def _r128(x):
    return (x + 127) // 128 * 128


_mlp_dtype = "fp32"


def set_mlp_dtype(dtype: str):
    """Compute type of the module-level MLP stacks (`Actor` / `Critic` / `bcqPerturbator` / `bcqGenerator` called directly, and
    therefore of `bcq_update`): 'fp32' (default; exact-fp32 MFMA, the parity mode) or 'bf16' (bf16 MFMA with fp32 accumulation:
    bf16 copies of the weights kept per weight version, activations and backward tensors in bf16, weight / bias gradients,
    outputs and every optimizer in fp32).  The fused DDPG / TD3 engine has its own switch (`fused.set_defaults(dtype=...)`)."""
    global _mlp_dtype
    if dtype not in ("fp32", "bf16"):
        raise ValueError(dtype)
    _mlp_dtype = dtype


def _args(M, N, dtype=None):
    a = L.GemmArgs()
    a.dtype = L.F32 if dtype is None else dtype
    a.M, a.N = M, N
    a.dx_scale = 1.0
    a.dw_splits = 1
    return a


def _fwd(x, K, w, bias, out, ldc, N, relu, mask, addend=None, yref=None, scale=1.0, dtype=None, add_row_div=0, c_f32=1):
    a = _args(x.shape[0], N, dtype)
    a.A[0], a.B[0], a.lda[0], a.ldb[0], a.K[0] = x.data_ptr(), w.data_ptr(), x.stride(0), w.stride(0), K
    a.C, a.ldc, a.c_f32 = out.data_ptr(), ldc, c_f32
    a.bias, a.relu = (bias.data_ptr() if bias is not None else None), int(relu)
    if addend is not None:      # added before the relu (gemm.hip epilogue_fwd)
        a.addend, a.ld_add, a.add_clip = addend.data_ptr(), addend.stride(0), float("inf")
        a.add_row_div = add_row_div
    if yref is not None:
        a.yref, a.ldy, a.dx_scale = yref.data_ptr(), yref.stride(0), scale
    if mask is not None:
        a.mask_mode, a.mask, a.ld_mask = L.MASK_EXTERNAL, mask.data_ptr(), mask.stride(0)
    if K >= 32768 and x.shape[0] * N <= (1 << 20):
        # a catalogue-long contraction with a small output: scratch for the library's split-K (recnn_gemm_args::ws)
        ws = _splitk_scratch(x.device, 8 * x.shape[0] * N)
        a.ws, a.ws_bytes = ws.data_ptr(), ws.numel() * 4
    L.call("recnn_gemm_fwd", C.byref(a), L.current_stream())


_splitk_ws = {}          # (device type, index, stream) -> [tensor, handed out during a stream capture?]
_splitk_retired = []     # outgrown buffers a captured graph may still replay into: alive for the process, nothing else is kept
_SPLITK_MAX_STREAMS = 8  # eager scratch buffers kept (least recently used first out); captured ones are never dropped


def _splitk_scratch(device, floats):
    """fp32 scratch of at least `floats` elements on `device` for the library's split-K, one per (device, STREAM): products issued on
    different streams never share a buffer.  A buffer that was handed out DURING A STREAM CAPTURE has its pointer baked into a
    graph (GraphedUpdate) that may replay into it for the life of the process: when it has to grow it is retired, not freed, and its
    stream's entry is never evicted.  Buffers only ever used eagerly are plain cache entries: replaced when they grow, and at most
    _SPLITK_MAX_STREAMS of them are kept (every stream used to pin its buffer, and the outgrown ones, forever)."""
    key = (device.type, device.index, int(torch.cuda.current_stream(device).cuda_stream))
    capturing = torch.cuda.is_current_stream_capturing()
    ent = _splitk_ws.pop(key, None)               # (re-inserted below: dict order = recency)
    if ent is None or ent[0].numel() < floats:
        if ent is not None and ent[1]:
            _splitk_retired.append(ent[0])
        ent = [torch.empty(floats, dtype=torch.float32, device=device), False]
    ent[1] = ent[1] or capturing
    _splitk_ws[key] = ent
    eager = [k for k, e in _splitk_ws.items() if not e[1]]
    for k in eager[:max(0, len(eager) - _SPLITK_MAX_STREAMS)]:
        del _splitk_ws[k]
    return ent[0]


def _dx(dz, Kc, w, N, out, yref, scale, colsum, dtype=None, c_f32=1):
    a = _args(dz.shape[0], N, dtype)
    a.A[0], a.B[0], a.lda[0], a.ldb[0], a.K[0] = dz.data_ptr(), w.data_ptr(), dz.stride(0), w.stride(0), Kc
    a.C, a.ldc, a.c_f32 = out.data_ptr(), out.stride(0), c_f32
    if yref is not None:
        a.yref, a.ldy, a.dx_scale = yref.data_ptr(), yref.stride(0), scale
    if colsum is not None:
        a.colsum = colsum.data_ptr()
    L.call("recnn_gemm_dx", C.byref(a), L.current_stream())


def _dw(dz, M, x, N, out, dtype=None):
    a = _args(M, N, dtype)
    a.A[0], a.B[0], a.lda[0], a.ldb[0], a.K[0] = dz.data_ptr(), x.data_ptr(), dz.stride(0), x.stride(0), dz.shape[0]
    a.C, a.ldc = out.data_ptr(), N
    a.dw_splits, a.dw_slab_stride, a.dw_valid_cols, a.dw_col_rot = 1, M * N, N, 0
    L.call("recnn_gemm_dw", C.byref(a), L.current_stream())


class MLPFunction(torch.autograd.Function):
    @classmethod
    def apply(cls, *args):
        # autograd runs Function.forward with grad mode OFF whatever the caller's mode is, so the caller's mode (does a backward
        # pass through this call exist at all?) is captured here and handed in as the last argument
        args = tuple(args) + (None,) * (11 - len(args))
        return super().apply(*args, torch.is_grad_enabled())

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, w3, b3, train, seed, masks=None, addend1=None, grad_mode=True):
        if not x.is_cuda:
            raise L.RecnnHipError("recnn_amd networks run on the GPU only (no CPU fallback): move the module and its "
                                  "inputs to 'cuda'")
        B, K = x.shape
        H, O = w1.shape[0], w3.shape[0]
        dev = x.device
        if _mlp_dtype == "bf16" and K < 4096:
            return MLPFunction._forward_bf16(ctx, x, w1, b1, w2, b2, w3, b3, train, seed, masks, addend1)
        Kp, Hp, Op = _r64(K), _r64(H), _r64(O)
        # catalogue-wide layer 1 (the critic over action distributions) in bf16 mode: the fp32 operands are only
        # materialised when a backward pass will need them
        big16 = _catalogue_dtype == "bf16" and K >= 4096
        # (needs_input_grad reflects requires_grad of the inputs even under torch.no_grad(): the reward / target forwards of
        #  reinforce_update run there and must not pay an 830 MB fp32 copy of W1 they never read)
        lean = big16 and not (grad_mode and any(ctx.needs_input_grad))
        xp = None if lean else _pad(x, B, Kp)
        if lean:
            w1p = None
        elif w1.numel() >= (1 << 22) and w1.is_leaf:
            w1p = _derived_of(w1, "padded", lambda w: _pad(w, Hp, Kp))      # 100k-wide rows are not 16-byte aligned as stored
        else:
            w1p = _pad(w1, Hp, Kp)
        w2p, w3p = _pad(w2, Hp, Hp), _pad(w3, Op, Hp)
        h1 = torch.zeros(B, Hp, device=dev)
        h2 = torch.zeros(B, Hp, device=dev)
        out = torch.empty(B, O, device=dev)
        m1 = m2 = None
        if train and masks is not None:
            m1, m2 = (m.to(device=dev, dtype=torch.uint8).contiguous() for m in masks)
        elif train:
            m1 = torch.empty(B, H, dtype=torch.uint8, device=dev)
            m2 = torch.empty(B, H, dtype=torch.uint8, device=dev)
            _dump_masks(seed, B, H, m1, m2, L.current_stream())
        b1c, b2c, b3c = b1.detach().float().contiguous(), b2.detach().float().contiguous(), b3.detach().float().contiguous()
        add1 = None if addend1 is None else addend1.detach().float().contiguous()
        if big16:
            # the critic over [state | action distribution]: layer 1 contracts over the catalogue
            K16 = _r128(K)
            x16 = torch.zeros(B, K16, dtype=torch.bfloat16, device=dev)
            x16[:, :K] = x
            def shadow(w):
                t = torch.zeros(Hp, K16, dtype=torch.bfloat16, device=dev)
                t[:H, :K] = w
                return t
            w16 = _derived_of(w1, "bf16_padded", shadow)
            _fwd(x16, K16, w16, b1c, h1, Hp, H, True, m1, addend=add1, dtype=L.BF16)
        else:
            _fwd(xp, Kp, w1p, b1c, h1, Hp, H, True, m1, addend=add1)
        _fwd(h1, Hp, w2p, b2c, h2, Hp, H, True, m2)
        _fwd(h2, Hp, w3p, b3c, out, O, O, False, None)
        ctx.save_for_backward(xp, h1, h2, w1p, w2p, w3p)
        ctx.dims = (B, K, H, O, Kp, Hp, Op)
        ctx.train = bool(train)
        ctx.bf16 = False
        return out

    # ---- bf16 compute mode (set_mlp_dtype): the same three GEMM + epilogue launches on bf16 operands
    @staticmethod
    def _shadow16(w, rows, cols):
        def build(t):
            out = torch.zeros(rows, cols, dtype=torch.bfloat16, device=t.device)
            out[: t.shape[0], : t.shape[1]] = t
            return out
        return _derived_of(w, f"bf16_{rows}x{cols}", build) if w.is_leaf else build(w.detach())

    @staticmethod
    def _buf16(rows, cols, valid, dev):
        """bf16 [rows, cols] whose columns >= valid are zero and whose first `valid` columns the caller's kernel writes: only the
        padding is filled (whole-buffer torch.zeros calls were 17 % of the BCQ step's kernel time, profiles/r03_bcq_*)."""
        t = torch.empty(rows, cols, dtype=torch.bfloat16, device=dev)
        if cols != valid:
            t[:, valid:].zero_()
        return t

    @staticmethod
    def _forward_bf16(ctx, x, w1, b1, w2, b2, w3, b3, train, seed, masks, addend1):
        B, K = x.shape
        H, O = w1.shape[0], w3.shape[0]
        Kp, Hp, Op = _r128(K), _r128(H), _r64(O)
        dev = x.device
        x16 = MLPFunction._buf16(B, Kp, K, dev)
        x16[:, :K] = x.detach()
        w1s, w2s, w3s = MLPFunction._shadow16(w1, Hp, Kp), MLPFunction._shadow16(w2, Hp, Hp), MLPFunction._shadow16(w3, Op, Hp)
        m1 = m2 = None
        if train and masks is not None:
            m1, m2 = (m.to(device=dev, dtype=torch.uint8).contiguous() for m in masks)
        elif train:
            m1 = torch.empty(B, H, dtype=torch.uint8, device=dev)
            m2 = torch.empty(B, H, dtype=torch.uint8, device=dev)
            _dump_masks(seed, B, H, m1, m2, L.current_stream())
        f = lambda t: t.detach().float().contiguous()
        add1 = None if addend1 is None else f(addend1)
        h1 = MLPFunction._buf16(B, Hp, H, dev)
        h2 = MLPFunction._buf16(B, Hp, H, dev)
        out = torch.empty(B, O, device=dev)
        _fwd(x16, Kp, w1s, f(b1), h1, Hp, H, True, m1, addend=add1, dtype=L.BF16, c_f32=0)
        _fwd(h1, Hp, w2s, f(b2), h2, Hp, H, True, m2, dtype=L.BF16, c_f32=0)
        _fwd(h2, Hp, w3s, f(b3), out, O, O, False, None, dtype=L.BF16)
        ctx.save_for_backward(x16, h1, h2, w1s, w2s, w3s)
        ctx.dims = (B, K, H, O, Kp, Hp, Op)
        ctx.train = bool(train)
        ctx.bf16 = True
        return out

    @staticmethod
    def _backward_bf16(ctx, dout):
        x16, h1, h2, w1s, w2s, w3s = ctx.saved_tensors
        B, K, H, O, Kp, Hp, Op = ctx.dims
        dev = dout.device
        scale = 2.0 if ctx.train else 1.0
        tiles = (B + 31) // 32
        need = ctx.needs_input_grad
        d16 = MLPFunction._buf16(B, Op, O, dev)
        d16[:, :O] = dout
        gw3 = gb3 = gw2 = gw1 = None
        if need[5]:
            gw3 = torch.empty(O, H, device=dev)
            _dw(d16, O, h2, H, gw3, dtype=L.BF16)
        if need[6]:
            gb3 = dout.float().sum(0)
        dz2 = MLPFunction._buf16(B, Hp, H, dev)
        cs2 = torch.empty(tiles, H, device=dev)
        _dx(d16, Op, w3s, H, dz2, h2, scale, cs2, dtype=L.BF16, c_f32=0)
        if need[3]:
            gw2 = torch.empty(H, H, device=dev)
            _dw(dz2, H, h1, H, gw2, dtype=L.BF16)
        dz1 = MLPFunction._buf16(B, Hp, H, dev)
        cs1 = torch.empty(tiles, H, device=dev)
        _dx(dz2, Hp, w2s, H, dz1, h1, scale, cs1, dtype=L.BF16, c_f32=0)
        if need[1]:
            gw1 = torch.empty(H, K, device=dev)
            _dw(dz1, H, x16, K, gw1, dtype=L.BF16)
        gx = None
        if need[0]:
            gx = torch.empty(B, K, device=dev)
            _dx(dz1, Hp, w1s, K, gx, None, 1.0, None, dtype=L.BF16)
        gadd = dz1[:, :H].float() if need[10] else None
        return (gx, gw1, cs1.sum(0) if need[2] else None, gw2, cs2.sum(0) if need[4] else None, gw3, gb3, None, None, None,
                gadd, None)

    @staticmethod
    def backward(ctx, dout):
        if ctx.bf16:
            return MLPFunction._backward_bf16(ctx, dout)
        xp, h1, h2, w1p, w2p, w3p = ctx.saved_tensors
        B, K, H, O, Kp, Hp, Op = ctx.dims
        dev = dout.device
        scale = 2.0 if ctx.train else 1.0
        tiles = (B + 31) // 32
        need = ctx.needs_input_grad      # frozen weights (a critic scoring a policy's action) skip their dW GEMMs
        doutp = _pad(dout, B, Op)
        gw3 = gb3 = gw2 = gw1 = None
        if need[5]:
            gw3 = torch.empty(O, H, device=dev)
            _dw(doutp, O, h2, H, gw3)
        if need[6]:
            gb3 = dout.sum(0)
        dz2 = torch.zeros(B, Hp, device=dev)
        cs2 = torch.empty(tiles, H, device=dev)
        _dx(doutp, Op, w3p, H, dz2, h2, scale, cs2)
        if need[3]:
            gw2 = torch.empty(H, H, device=dev)
            _dw(dz2, H, h1, H, gw2)
        dz1 = torch.zeros(B, Hp, device=dev)
        cs1 = torch.empty(tiles, H, device=dev)
        _dx(dz2, Hp, w2p, H, dz1, h1, scale, cs1)
        if need[1]:
            gw1 = torch.empty(H, K, device=dev)
            _dw(dz1, H, xp, K, gw1)
        gx = None
        if need[0]:
            gx = torch.empty(B, K, device=dev)
            _dx(dz1, Hp, w1p, K, gx, None, 1.0, None)
        gadd = dz1[:, :H] if need[10] else None     # d/d addend1 = dZ1
        return (gx, gw1, cs1.sum(0) if need[2] else None, gw2, cs2.sum(0) if need[4] else None, gw3, gb3, None, None, None,
                gadd, None)


def mlp(x, module, train: bool):
    """`module.forced_masks` (a list of (m1, m2) uint8 keep-mask pairs, consumed first-in first-out) replaces the hash masks
    of the next train-mode calls: replaying logged masks, and the parity tests."""
    return MLPFunction.apply(x.float(), module.linear1.weight, module.linear1.bias, module.linear2.weight,
                             module.linear2.bias, module.linear3.weight, module.linear3.bias, train, torch.initial_seed(),
                             _take_forced_masks(module, train), None)


def mlp_frozen(x, module, train: bool):
    """`mlp` with the module's weights taken as constants: the gradient reaches `x` only (a critic scoring the action of the
    policy being trained -- its own dW / db GEMMs would be thrown away by the next zero_grad)."""
    d = lambda t: t.detach()
    return MLPFunction.apply(x.float(), d(module.linear1.weight), d(module.linear1.bias), d(module.linear2.weight),
                             d(module.linear2.bias), d(module.linear3.weight), d(module.linear3.bias), train,
                             torch.initial_seed(), _take_forced_masks(module, train), None)


def mlp3(x, l1, l2, w3, b3):
    """relu(l1) -> relu(l2) -> x W3^T + b3 without dropout (the VAE encoder / decoder stacks of bcqGenerator)."""
    return MLPFunction.apply(x.float(), l1.weight, l1.bias, l2.weight, l2.bias, w3, b3, False, 0, None, None)


def mlp_candidates(state, x, n, w1, b1, w2, b2, w3, b3):
    """MLP([repeat_interleave(state, n, 0) | x]) for `x` holding n consecutive candidate rows per state row -- relu, relu,
    linear, no dropout, no gradient (the target side of BCQ's critic step, recnn/nn/update/bcq.py:98-106, where the
    reference materialises the repeated states and multiplies them n times over).  Layer 1 is split along its input:
    the state part  S1 = state W1[:, :S]^T + b1  is computed ONCE per state row, the candidate part adds it back per row
    through the GEMM epilogue (`add_row_div`: output row m reads S1 row m / n) before the relu.  10 candidates per state at
    state 1290 / candidate 128..512 wide: 4..8 x fewer layer-1 FLOPs and no [B n, 1290] operand in memory."""
    if not state.is_cuda:
        raise L.RecnnHipError("recnn_amd networks run on the GPU only (no CPU fallback)")
    with torch.no_grad():
        B, S = state.shape
        R, Kx = x.shape
        if R != B * n or w1.shape[1] != S + Kx:
            raise ValueError(f"mlp_candidates: {R} candidate rows for {B} states x {n}, layer 1 takes {w1.shape[1]} inputs")
        H, O = w1.shape[0], w3.shape[0]
        Sp, Kp, Hp, Op = _r64(S), _r64(Kx), _r64(H), _r64(O)
        dev = state.device
        f = lambda t: t.detach().float().contiguous()
        if _mlp_dtype == "bf16":
            Sp, Kp, Hp = _r128(S), _r128(Kx), _r128(H)
            buf16 = MLPFunction._buf16

            def to16(t, rows, cols):       # only the padding is filled (whole-buffer zeros + copies were 11 % of the graphed BCQ step)
                o = buf16(rows, cols, t.shape[1], dev)
                if rows != t.shape[0]:
                    o[t.shape[0]:].zero_()
                o[: t.shape[0], : t.shape[1]] = t.detach()
                return o

            def w16(part, lo, hi, cols):   # bf16 copy of W1's columns [lo, hi), kept per weight version when W1 is a leaf
                build = lambda w: to16(w[:, lo:hi], Hp, cols)
                return _derived_of(w1, f"bf16_cols_{lo}_{hi}_{Hp}x{cols}", build) if w1.is_leaf else build(w1)
            s1 = torch.empty(B, Hp, device=dev)                 # the shared state part stays fp32 (it is an epilogue addend)
            if Hp != H:
                s1[:, H:].zero_()
            _fwd(to16(state, B, Sp), Sp, w16("state", 0, S, Sp), f(b1), s1, Hp, H, False, None, dtype=L.BF16)
            h1 = buf16(R, Hp, H, dev)
            _fwd(to16(x, R, Kp), Kp, w16("cand", S, S + Kx, Kp), None, h1, Hp, H, True, None, addend=s1, add_row_div=n, dtype=L.BF16, c_f32=0)
            h2 = buf16(R, Hp, H, dev)
            _fwd(h1, Hp, MLPFunction._shadow16(w2, Hp, Hp), f(b2), h2, Hp, H, True, None, dtype=L.BF16, c_f32=0)
            out = torch.empty(R, O, device=dev)
            _fwd(h2, Hp, MLPFunction._shadow16(w3, Op, Hp), f(b3), out, O, O, False, None, dtype=L.BF16)
            return out
        s1 = torch.zeros(B, Hp, device=dev)
        _fwd(_pad(state, B, Sp), Sp, _pad(w1[:, :S], Hp, Sp), f(b1), s1, Hp, H, False, None)
        h1 = torch.zeros(R, Hp, device=dev)
        _fwd(_pad(x, R, Kp), Kp, _pad(w1[:, S:], Hp, Kp), None, h1, Hp, H, True, None, addend=s1, add_row_div=n)
        h2 = torch.zeros(R, Hp, device=dev)
        _fwd(h1, Hp, _pad(w2, Hp, Hp), f(b2), h2, Hp, H, True, None)
        out = torch.empty(R, O, device=dev)
        _fwd(h2, Hp, _pad(w3, Op, Hp), f(b3), out, O, O, False, None)
        return out


# ------------------------------------------------------------------------------------------------------------------------
# Conditional VAE of BCQ: latent layer and loss as autograd nodes over csrc/vae.hip.
# Replaces recnn/nn/models.py:271-277 and recnn/nn/update/bcq.py:78-81 (and the graph torch would record for them).

class VaeLatentFunction(torch.autograd.Function):
    """(z, std) = latent(ml, eps): ml = [mean | raw log_std]; std = exp(clamp(raw, -4, 15)); z = mean + std * eps."""

    @staticmethod
    def forward(ctx, ml, eps):
        if not ml.is_cuda:
            raise L.RecnnHipError("recnn_amd VAE latent layer: needs GPU tensors (no CPU fallback)")
        ml = ml.float().contiguous()
        eps = eps.to(device=ml.device, dtype=torch.float32).contiguous()
        B, L2 = ml.shape
        Ld = L2 // 2
        z = torch.empty(B, Ld, device=ml.device)
        std = torch.empty(B, Ld, device=ml.device)
        L.call("recnn_vae_latent_fwd", L.ptr(ml), ml.stride(0), L.ptr(eps), eps.stride(0), B, Ld, L.ptr(z), z.stride(0),
               L.ptr(std), std.stride(0), L.current_stream())
        ctx.save_for_backward(ml, eps, std)
        return z, std

    @staticmethod
    def backward(ctx, dz, dstd):
        ml, eps, std = ctx.saved_tensors
        B, L2 = ml.shape
        Ld = L2 // 2
        rows = lambda t: None if t is None else (t.float() if t.stride(1) == 1 else t.float().contiguous())
        dz, dstd = rows(dz), rows(dstd)        # dz is usually the latent columns of the decoder's dX: strided rows
        dml = torch.empty_like(ml)
        L.call("recnn_vae_latent_bwd", L.ptr(ml), ml.stride(0), L.ptr(eps), eps.stride(0), L.ptr(std), std.stride(0),
               L.ptr(dz), 0 if dz is None else dz.stride(0), None, 0, L.ptr(dstd), 0 if dstd is None else dstd.stride(0),
               B, Ld, L.ptr(dml), dml.stride(0), L.current_stream())
        return dml, None


def vae_latent(ml, eps):
    return VaeLatentFunction.apply(ml, eps)


class VaeLossFunction(torch.autograd.Function):
    """out3 = (mse(recon, action), KL, mse + kl_weight * KL), KL = -0.5 mean(1 + log(std^2) - mean^2 - std^2)."""

    @staticmethod
    def forward(ctx, recon, action, mean, std, kl_weight):
        if not recon.is_cuda:
            raise L.RecnnHipError("recnn_amd VAE loss: needs GPU tensors (no CPU fallback)")
        recon, action = recon.float(), action.detach().float()
        mean, std = mean.float(), std.float()
        for t in (recon, action, mean, std):
            if t.stride(1) != 1:
                raise L.RecnnHipError("recnn_amd VAE loss: rows must be contiguous")
        B, A = recon.shape
        Ld = mean.shape[1]
        out = torch.empty(3, device=recon.device)
        scratch = torch.empty(512, device=recon.device)
        L.call("recnn_vae_loss_fwd", L.ptr(recon), recon.stride(0), L.ptr(action), action.stride(0), L.ptr(mean), mean.stride(0),
               L.ptr(std), std.stride(0), B, A, Ld, float(kl_weight), L.ptr(out), L.ptr(scratch), L.current_stream())
        ctx.save_for_backward(recon, action, mean, std)
        ctx.kl_weight = float(kl_weight)
        return out

    @staticmethod
    def backward(ctx, g):
        recon, action, mean, std = ctx.saved_tensors
        B, A = recon.shape
        Ld = mean.shape[1]
        g = g.float().contiguous()
        d_recon = torch.empty(B, A, device=recon.device)
        d_mean = torch.empty(B, Ld, device=recon.device)
        d_std = torch.empty(B, Ld, device=recon.device)
        L.call("recnn_vae_loss_bwd", L.ptr(recon), recon.stride(0), L.ptr(action), action.stride(0), L.ptr(mean), mean.stride(0),
               L.ptr(std), std.stride(0), B, A, Ld, L.ptr(g), ctx.kl_weight, L.ptr(d_recon), A, L.ptr(d_mean), Ld,
               L.ptr(d_std), Ld, L.current_stream())
        return d_recon, None, d_mean, d_std, None


def vae_loss(recon, action, mean, std, kl_weight=0.5):
    """float[3] = (reconstruction loss, KL loss, reconstruction + kl_weight * KL)  (bcq.py:78-81)."""
    return VaeLossFunction.apply(recon, action, mean, std, kl_weight)


def _take_forced_masks(module, train):
    forced = getattr(module, "forced_masks", None)
    return forced.pop(0) if (train and forced) else None


# ------------------------------------------------------------------------------------------------------------------------
# Categorical policy head of REINFORCE: DiscreteActor = linear1 -> relu -> linear2 -> softmax, plus
# Categorical sampling / log-prob, as ONE autograd node over the HIP kernels (csrc/gemm.hip, csrc/policy.hip).
# Replaces recnn/nn/models.py:95-111 and the autograd graph torch would record for it.

def _r4(x):
    return (x + 3) // 4 * 4


def categorical(probs, actions=None, seed=None):
    """Categorical(probs).sample() and .log_prob(action) of torch.distributions (probs / probs.sum, clamp to [eps, 1-eps], log)
    over rows of an (unnormalised) probability matrix, no gradient.  actions=None draws them.  Returns (actions, log_prob)."""
    if not probs.is_cuda:
        raise L.RecnnHipError("recnn_amd categorical: needs a GPU tensor (no CPU fallback)")
    p = probs.detach()
    B, N = p.shape
    if p.dtype != torch.float32 or p.stride(1) != 1 or p.stride(0) % 4 or p.data_ptr() % 16:
        q = torch.zeros(B, _r4(N), dtype=torch.float32, device=p.device)
        q[:, :N] = p
        p = q
    act = torch.empty(B, dtype=torch.int64, device=p.device) if actions is None else actions.to(torch.int64).contiguous()
    lp = torch.empty(B, dtype=torch.float32, device=p.device)
    flags = L.CAT_SAMPLE if actions is None else 0
    seed = torch.initial_seed() if seed is None else seed
    L.call("recnn_categorical_rows", L.ptr(p), p.stride(0), B, N, flags, seed & 0xFFFFFFFF, next(_call_counter) & 0x7FFFFFFF,
           L.ptr(act), L.ptr(lp), None, L.current_stream())
    return act, lp


class DiscretePolicyFunction(torch.autograd.Function):
    """(probs, action, log_prob) = head(x): h = relu(x W1^T + b1), p = softmax(h W2^T + b2), action ~ p (or given),
    log_prob = log(clamp(p[action] / sum p)).  Backward takes d log_prob (REINFORCE) and / or d probs."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, actions, sample, seed):
        if not x.is_cuda:
            raise L.RecnnHipError("recnn_amd networks run on the GPU only (no CPU fallback): move the module and its "
                                  "inputs to 'cuda'")
        ctx.set_materialize_grads(False)
        B, K = x.shape
        H, N = w1.shape[0], w2.shape[0]
        bf16 = _catalogue_dtype == "bf16"
        Kp, Hp, ldn = _r64(K), (_r128(H) if bf16 else _r64(H)), (_r128(N) if bf16 else _r64(N))
        dev = x.device
        xp = _pad(x, B, Kp)
        w1p = _pad(w1, Hp, Kp)
        h = torch.zeros(B, Hp, device=dev)
        buf = torch.empty(B, ldn, device=dev)
        if ldn != N:
            buf[:, N:].zero_()
        _fwd(xp, Kp, w1p, b1.detach().float().contiguous(), h, Hp, H, True, None)
        if bf16:
            def shadow(w):
                t = torch.zeros(N, Hp, dtype=torch.bfloat16, device=dev)
                t[:, :H] = w
                return t
            w2p = _derived_of(w2, "bf16", shadow)
            h16 = h.to(torch.bfloat16)
            _fwd(h16, Hp, w2p, b2.detach().float().contiguous(), buf, ldn, N, False, None, dtype=L.BF16)
        else:
            w2p = _pad(w2, _r4(N), Hp)     # no copy for the usual shapes (N % 4 == 0, H % 64 == 0)
            _fwd(h, Hp, w2p, b2.detach().float().contiguous(), buf, ldn, N, False, None)
        stat = torch.empty(B, 4, device=dev)
        lp = torch.zeros(B, device=dev)
        flags = L.CAT_SOFTMAX
        if sample:
            act = torch.empty(B, dtype=torch.int64, device=dev)
            flags |= L.CAT_SAMPLE
        elif actions is not None:
            act = actions.to(device=dev, dtype=torch.int64).contiguous()
        else:
            act = None
        L.call("recnn_categorical_rows", L.ptr(buf), ldn, B, N, flags, seed & 0xFFFFFFFF, next(_call_counter) & 0x7FFFFFFF,
               L.ptr(act), L.ptr(lp), L.ptr(stat), L.current_stream())
        probs = buf[:, :N]
        act_out = act if act is not None else torch.full((B,), -1, dtype=torch.int64, device=dev)
        ctx.save_for_backward(xp, h, w1p, w2p, probs, act_out, stat)
        ctx.w2_param = w2
        ctx.bf16 = bf16
        ctx.dims = (B, K, H, N, Kp, Hp, ldn)
        ctx.has_action = act is not None
        ctx.mark_non_differentiable(act_out)
        return probs, act_out, lp

    @staticmethod
    def backward(ctx, dprobs, _dact, dlp):
        xp, h, w1p, w2p, probs, act, stat = ctx.saved_tensors
        B, K, H, N, Kp, Hp, ldn = ctx.dims
        dev = probs.device
        s = L.current_stream()
        if dprobs is None and (dlp is None or not ctx.has_action):
            return (None,) * 8
        bf16 = ctx.bf16 and dprobs is None      # (a gradient through the probabilities themselves stays on the fp32 path)
        dlog = torch.empty(B, ldn, device=dev, dtype=torch.bfloat16 if bf16 else torch.float32)
        if ldn != _r4(N):
            dlog[:, _r4(N):].zero_()
        acc = 0
        if dprobs is not None:
            dpr = dprobs.float().contiguous()
            L.call("recnn_softmax_bwd", L.ptr(probs), probs.stride(0), B, N, L.ptr(dpr), dpr.stride(0), L.ptr(dlog), ldn, s)
            acc = L.LPB_ACCUMULATE
        if bf16:
            acc |= L.LPB_BF16
        g = None
        if dlp is not None and ctx.has_action:
            g = dlp.float().contiguous()
        if not ctx.has_action:
            act = torch.zeros(B, dtype=torch.int64, device=dev)
        gb2 = torch.empty(N, device=dev)
        scratch = torch.empty((B + 31) // 32, _r4(N), device=dev)
        L.call("recnn_logprob_bwd", L.ptr(probs), probs.stride(0), B, N, L.ptr(act), L.ptr(g), L.ptr(stat), L.ptr(dlog), ldn, acc,
               L.ptr(gb2), L.ptr(scratch), s)
        del scratch
        gdt = L.BF16 if bf16 else None
        h_op = h.to(torch.bfloat16) if bf16 else h       # [rows, hidden]: small next to the catalogue operands
        need_w2 = bool(ctx.needs_input_grad[3])          # (a frozen head -- requires_grad False -- gets no gradient, deferred or not)
        defer = need_w2 and _episode["on"] and dprobs is None and ctx.w2_param.is_leaf
        gw2 = None                                       # deferred: ONE dW over the whole episode's rows at the end of the pass (episode_backward)
        if need_w2 and not defer:
            gw2 = torch.empty(N, H, device=dev)
            _dw(dlog, N, h_op, H, gw2, dtype=gdt)
        dz1 = torch.zeros(B, Hp, device=dev)
        # dZ1 = (dlogits W2) * [h > 0]: contraction over the catalogue.  W2 is [n_items, hidden], k-strided for this product;
        # its transpose (made once per weight version, shared by the backward passes of a whole episode) puts the
        # contraction on the contiguous axis, so the LDS-DMA forward kernel runs it (measured 4.2 -> 1.3 ms at 256 x 100k x 2048)
        def transposed(w):
            return transposed_rows(w, ldn, bf16)
        w2t = _derived_of(ctx.w2_param, "transposed_bf16" if bf16 else "transposed", transposed)
        _fwd(dlog, ldn, w2t, None, dz1, Hp, H, False, None, yref=h_op, scale=1.0, dtype=gdt)
        if defer:     # grouped by (parameter, operand type): set_catalogue_dtype may change between the steps of an episode
            _episode["pending"].setdefault((id(ctx.w2_param), gdt), (ctx.w2_param, gdt, []))[2].append((dlog, h_op))
        del dlog
        gb1 = dz1[:, :H].sum(0)
        gw1 = torch.empty(H, K, device=dev)
        _dw(dz1, H, xp, K, gw1)
        gx = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty(B, K, device=dev)
            _dx(dz1, Hp, w1p, K, gx, None, 1.0, None)
        return gx, gw1, gb1, gw2, gb2, None, None, None


def transposed_rows(w, ldt, bf16=False):
    """[cols, ldt] (fp32 or bf16, padding columns zero) with out[c, r] = w[r, c], by the tiled HIP transpose (recnn_transpose_rows)."""
    w = w.detach().float().contiguous()
    R, Cc = w.shape
    t = torch.zeros(Cc, ldt, device=w.device, dtype=torch.bfloat16 if bf16 else torch.float32)
    L.call("recnn_transpose_rows", L.ptr(w), w.stride(0), R, Cc, L.ptr(t), ldt, int(bool(bf16)), L.current_stream())
    return t


# ---- the policy head's weight gradient over an EPISODE.  ChooseREINFORCE's loss is a sum over the `policy_step` steps of the
# episode, each term a separate autograd node of DiscretePolicyFunction: left to autograd, every node computes its own
# [n_items, hidden] gradient (a dW launch that writes 819 MB at 100k x 2048) and AccumulateGrad adds them one by one (nine 819 MB
# add_ passes).  Inside `episode_backward()` the nodes keep their d logits instead and ONE dW GEMM over all the episode's rows
# (k = steps x batch) writes the gradient once; it is added to `.grad` like autograd would.
_episode = {"on": False, "pending": {}}


class episode_backward:
    """`with episode_backward(): loss.backward()` -- see above.  Only nodes whose log-prob (not the probabilities) carries the
    gradient take part; everything else in the graph is untouched."""

    def __enter__(self):
        if _episode["on"]:
            raise RuntimeError("episode_backward is not re-entrant")
        _episode["on"], _episode["pending"] = True, {}
        return self

    def __exit__(self, exc_type, exc, tb):
        pending, _episode["pending"], _episode["on"] = _episode["pending"], {}, False
        if exc_type is not None:
            return False
        def gather(terms, i):
            # the terms' tensors into one [rows, ld] operand, each released as soon as it is copied (torch.cat would hold the
            # episode twice: ~2 x 1 GB of d logits at 10 x 256 x 100k)
            if len(terms) == 1:
                return terms[0][i]
            first = terms[0][i]
            out = torch.empty((sum(t[i].shape[0] for t in terms),) + tuple(first.shape[1:]), dtype=first.dtype, device=first.device)
            off = 0
            for k in range(len(terms)):
                t = terms[k][i]
                out[off:off + t.shape[0]].copy_(t)
                off += t.shape[0]
                terms[k] = tuple(None if j == i else terms[k][j] for j in range(2))
            return out
        for w2, gdt, terms in pending.values():
            N, H = w2.shape
            dlog = gather(terms, 0)
            h_op = gather(terms, 1)
            gw2 = torch.empty(N, H, device=w2.device)
            _dw(dlog, N, h_op, H, gw2, dtype=gdt)
            del dlog, h_op
            with torch.no_grad():
                if w2.grad is None:
                    w2.grad = gw2
                else:
                    w2.grad.add_(gw2)
        return False


def discrete_policy(x, module, actions=None, sample=False):
    """probs, action, log_prob of `module` (a DiscreteActor) for a batch of states."""
    return DiscretePolicyFunction.apply(x.float(), module.linear1.weight, module.linear1.bias, module.linear2.weight,
                                        module.linear2.bias, actions, bool(sample), torch.initial_seed())


def beta_train_forward(state, target, weight, bias):
    """The learned behaviour policy of the Top-K off-policy correction (`Beta` of the reference's notebook
    `examples/2. REINFORCE TopK Off Policy Correction/3. TopK Reinforce Off Policy Correction.ipynb`, cell 3):

        p    = Softmax()(Linear(state))                      # the probabilities it returns (detached)
        loss = CrossEntropyLoss()(p, action.argmax(1))       # NB: the cross entropy of the PROBABILITIES taken as logits

    Returns (p float32 [B, n_items] -- a view into a 16-byte aligned buffer --, loss (device scalar), dW [n_items, K], db [n_items]): every
    catalogue-wide pass runs in HIP kernels -- the logits GEMM, the two softmaxes (of the logits, and of p inside the loss:
    recnn_categorical_rows), d loss / d p = (softmax(p) - onehot) / B (recnn_logprob_bwd with g = -1 / B), the softmax backward
    (recnn_softmax_bwd) and the weight-gradient GEMM; the bias gradient is the column sum of d logits (recnn_colsum_rows).  No autograd graph is recorded (the notebook's forward steps its optimizer itself)."""
    if not state.is_cuda:
        raise L.RecnnHipError("recnn_amd Beta: needs GPU tensors (no CPU fallback)")
    x = state.detach().float()
    B, K = x.shape
    N = weight.shape[0]
    dev = x.device
    Kp, ldn = _r64(K), _r64(N)
    s = L.current_stream()
    xp = _pad(x, B, Kp)
    p = torch.empty(B, ldn, device=dev)
    if ldn != N:
        p[:, N:].zero_()
    if _catalogue_dtype == "bf16" and N >= 4096:
        # the logits GEMM is catalogue-sized ([B, 1290] x [n_items, 1290]^T): bf16 operands in the bf16 catalogue mode, like the policy
        # head's; the bf16 copy of the weight is kept current by the optimizer pass (shadow_target)
        K16 = _r128(K)
        x16 = torch.zeros(B, K16, dtype=torch.bfloat16, device=dev)
        x16[:, :K] = x

        def shadow(w):
            t = torch.zeros(_r4(N), K16, dtype=torch.bfloat16, device=dev)
            t[:N, :K] = w
            return t
        w16 = _derived_of(weight, "bf16_padded", shadow)
        _fwd(x16, K16, w16, bias.detach().float().contiguous(), p, ldn, N, False, None, dtype=L.BF16)
    else:
        wp = _derived_of(weight, "padded", lambda w: _pad(w.detach(), _r4(N), Kp)) if (K != Kp or N != _r4(N)) else weight.detach()
        _fwd(xp, Kp, wp, bias.detach().float().contiguous(), p, ldn, N, False, None)
    tgt = target.to(device=dev, dtype=torch.int64).contiguous()
    L.call("recnn_categorical_rows", L.ptr(p), ldn, B, N, L.CAT_SOFTMAX, 0, 0, None, None, None, s)          # p = softmax(logits)
    q = p.clone()
    stat = torch.empty(B, 4, device=dev)
    lp = torch.empty(B, device=dev)
    L.call("recnn_categorical_rows", L.ptr(q), ldn, B, N, L.CAT_SOFTMAX, 0, 0, L.ptr(tgt), L.ptr(lp), L.ptr(stat), s)   # q = softmax(p), lp = log q[target]
    loss = -lp.mean()
    g = torch.full((B,), -1.0 / B, device=dev)
    dprobs = torch.empty(B, ldn, device=dev)
    if ldn != _r4(N):
        dprobs[:, _r4(N):].zero_()
    L.call("recnn_logprob_bwd", L.ptr(q), ldn, B, N, L.ptr(tgt), L.ptr(g), L.ptr(stat), L.ptr(dprobs), ldn, 0, None, None, s)
    del q
    dlog = torch.empty(B, ldn, device=dev)
    if ldn != _r4(N):
        dlog[:, _r4(N):].zero_()
    L.call("recnn_softmax_bwd", L.ptr(p), ldn, B, N, L.ptr(dprobs), ldn, L.ptr(dlog), ldn, s)
    del dprobs
    gw = torch.empty(N, K, device=dev)
    if _catalogue_dtype == "bf16" and N >= 4096:
        # the weight gradient d logits^T x state ([n_items, 256] x [256, 1290]: 66 GFLOP at 100k items) on the bf16 LDS-DMA dW kernel like the
        # policy head's; the fp32 register-staged kernel it replaces was 17.5 % of the 100k bf16 step (profiles/r04_reinforce_100k_bf16_*)
        _dw(dlog.to(torch.bfloat16), N, x16, K, gw, dtype=L.BF16)
    else:
        _dw(dlog, N, xp, K, gw)
    gb = torch.empty(N, device=dev)
    L.call("recnn_colsum_rows", L.ptr(dlog), ldn, B, N, L.ptr(gb), s)      # sum over the batch rows of d logits
    return p[:, :N], loss, gw, gb


def onehot_rows(idx, n):
    """float[B, n] one-hot rows of int64 indices (recnn/data/utils.py:108-109: zeros + scatter_).  The result remembers
    its indices (`onehot_index`): a Critic reading it gathers the B weight columns instead of multiplying a [B, n] matrix
    of zeros (`mlp_onehot`)."""
    B = idx.numel()
    ld = _r4(n)
    idx = idx.to(torch.int64).contiguous()
    if B:
        lo, hi = torch.aminmax(idx)
        if int(lo) < 0 or int(hi) >= n:      # the reference's scatter_ raises for such ids (utils.py:108-109)
            raise IndexError(f"onehot_rows: index range [{int(lo)}, {int(hi)}] outside the catalogue [0, {n})")
    out = torch.empty(B, ld, dtype=torch.float32, device=idx.device)
    L.call("recnn_onehot_rows", L.ptr(idx), B, n, L.ptr(out), ld, L.current_stream())
    out = out if ld == n else out[:, :n]
    out.onehot_index = (idx, out._version)
    return out


def onehot_index_of(t):
    """The indices of a tensor made by `onehot_rows`, or None (also when it was written to since)."""
    tag = getattr(t, "onehot_index", None)
    if tag is None or tag[1] != t._version or tag[0].numel() != t.shape[0]:
        return None
    return tag[0]


class OneHotLayer1Function(torch.autograd.Function):
    """(W1[:, :S], W1[:, S + idx]^T) of a critic's first layer over [state | one-hot action].  Written as plain slicing +
    index_select, autograd answers with TWO catalogue-wide gradients for W1 -- zeros(W1) with the state block copied in, zeros(W1)
    with the columns scattered -- a clone and an add_ over them (at [2048, 101290]: two 830 MB fills, a copy and an add per step,
    1.2 ms); this node builds the ONE dense gradient the optimizer reads: a fill and two small writes."""

    @staticmethod
    def forward(ctx, w1, idx, n_state):
        ctx.save_for_backward(idx)
        ctx.meta = (tuple(w1.shape), int(n_state))
        return w1[:, :n_state].contiguous(), w1.index_select(1, idx + n_state).t().contiguous()

    @staticmethod
    def backward(ctx, g_state, g_cols):
        (idx,) = ctx.saved_tensors
        shape, n_state = ctx.meta
        ref = g_state if g_state is not None else g_cols
        g = torch.zeros(shape, dtype=torch.float32, device=ref.device)
        if g_state is not None:
            g[:, :n_state] = g_state
        if g_cols is not None:
            g.index_add_(1, idx + n_state, g_cols.t().float())
        return g, None, None


def mlp_onehot(state, idx, n_state, module, train: bool):
    """`mlp(cat([state, onehot(idx)], 1))` without the one-hot operand: layer 1 is state W1[:, :S]^T + W1[:, S + idx]^T + b1
    (the column gather enters the GEMM epilogue before the relu); autograd scatters the gathered columns' gradient back."""
    w1 = module.linear1.weight
    w_state, cols = OneHotLayer1Function.apply(w1, idx, n_state)           # [H, S], [B, H]
    return MLPFunction.apply(state.float(), w_state, module.linear1.bias, module.linear2.weight, module.linear2.bias,
                             module.linear3.weight, module.linear3.bias, train, torch.initial_seed(),
                             _take_forced_masks(module, train), cols)
