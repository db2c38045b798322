"""Optimizers.

`Adam`  -- torch.optim.Optimizer whose step() is the fused HIP kernel `recnn_adam_flat` (arithmetic of
           torch.optim.Adam: L2 weight decay, bias correction, eps outside the sqrt).  When both optimizers of
           an update are (this or torch's) plain Adam, `ddpg_update` / `td3_update` run Adam INSIDE the fused
           step engine (one launch per network, fused with the shadow refresh and the soft target update).
`Ranger`-- RAdam + Lookahead, the optimizer the reference builds by default (`torch_optimizer.Ranger(lr=1e-5,
           weight_decay=1e-2)`, recnn/nn/algo.py:84-89,139-147).  step() is the fused HIP kernel `recnn_ranger_flat`; inside
           `ddpg_update` / `td3_update` / `Algo.run` the same arithmetic runs in the engine's optimizer pass (fused with the
           gradient slab reduction, the shadow refresh and the soft target update).  `torch_optimizer` is an absent,
           un-pinned third-party package: the arithmetic follows its published algorithm from memory and is NOT verified
           against it ("parity unpinned").  What IS pinned: with weight_decay = 0 the RAdam part equals
           torch.optim.RAdam (same rectification term, same threshold, eps outside the sqrt), and Lookahead is three lines
           (tests/test_gpu_optim.py); `Ranger.reference_step` is the same algorithm in plain torch ops.
"""
import math

import ctypes as C

import torch

from . import _lib as L

__all__ = ["Adam", "Ranger", "adam_config", "fused_config"]


def _shadow_of(p):
    """(kind, ctypes reference to a recnn_shadow_out) for the cached compute-layout copy of a catalogue-sized 2-D parameter the
    kernel should rewrite in the same pass (recnn_amd.nn.functional.shadow_target), or (None, None).  Only worth it for big
    weights: below 4M elements the next forward's rebuild is noise."""
    if p.dim() != 2 or p.numel() < (1 << 22):
        return None, None
    from .nn import functional as Fh
    hit = Fh.shadow_target(p)
    if hit is None:
        return None, None
    kind, out = hit
    sh = L.ShadowOut()
    sh.dst, sh.cols, sh.ld, sh.bf16 = out.data_ptr(), int(p.shape[1]), int(out.stride(0)), int(out.dtype == torch.bfloat16)
    return kind, C.byref(sh)


def _shadow_done(p, kind):
    from .nn import functional as Fh
    Fh.shadow_written(p, kind)


class Adam(torch.optim.Optimizer):
    """torch.optim.Adam's arithmetic on the HIP kernel `recnn_adam_flat`.  capturable=True keeps the step count in device memory
    (one int32 per optimizer, advanced by a torch op after the parameters' launches): `step()` then has no host-side state that a
    captured graph would freeze (recnn_amd/nn/graphed.py); `state[p]['step']` still counts the host's calls."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, capturable=False):
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1) or weight_decay < 0:
            raise ValueError("invalid Adam hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.capturable = bool(capturable)
        self._step_dev = None

    def device_steps(self) -> int:
        """Optimizer steps taken as the device counts them (capturable mode; synchronises)."""
        return 0 if self._step_dev is None else int(self._step_dev.item())

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        stream = None
        stepped = False
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda:
                    raise L.RecnnHipError("recnn_amd.optim.Adam: parameters must live on the GPU (no CPU fallback)")
                if p.dtype != torch.float32 or not p.data.is_contiguous():
                    raise L.RecnnHipError("recnn_amd.optim.Adam: parameters must be contiguous float32")
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p.data)
                    st["exp_avg_sq"] = torch.zeros_like(p.data)
                st["step"] = int(st["step"]) + 1
                g = p.grad.data if p.grad.data.is_contiguous() else p.grad.data.contiguous()
                stream = stream or L.current_stream()
                if self.capturable:
                    if self._step_dev is None:
                        self._step_dev = torch.zeros(1, dtype=torch.int32, device=p.device)
                    stepped = True
                    L.call("recnn_adam_flat_at", L.ptr(p.data), L.ptr(g), L.ptr(st["exp_avg"]), L.ptr(st["exp_avg_sq"]), p.numel(),
                           float(group["lr"]), float(b1), float(b2), float(group["eps"]), float(group["weight_decay"]),
                           L.ptr(self._step_dev), 1, 1.0, stream)
                else:
                    kind, sh = _shadow_of(p)
                    L.call("recnn_adam_flat_shadow", L.ptr(p.data), L.ptr(g), L.ptr(st["exp_avg"]), L.ptr(st["exp_avg_sq"]), p.numel(),
                           float(group["lr"]), float(b1), float(b2), float(group["eps"]), float(group["weight_decay"]),
                           int(st["step"]), 1.0, sh, stream)
                torch.autograd.graph.increment_version(p)    # the kernel wrote p behind autograd's back
                if not self.capturable and kind is not None:
                    _shadow_done(p, kind)
        if stepped:
            self._step_dev.add_(1)
        return loss


def adam_config(opt):
    """Hyper-parameters if `opt` is an Adam the fused engine reproduces exactly, else None."""
    if opt is None or len(opt.param_groups) != 1:
        return None
    g = opt.param_groups[0]
    if type(opt) is Adam:
        pass
    elif type(opt) is torch.optim.Adam:
        if g.get("amsgrad") or g.get("maximize") or g.get("capturable") or g.get("differentiable"):
            return None
        if g.get("decoupled_weight_decay"):
            return None
        if isinstance(g["lr"], torch.Tensor):
            return None
    else:
        return None
    return dict(lr=float(g["lr"]), beta1=float(g["betas"][0]), beta2=float(g["betas"][1]), eps=float(g["eps"]),
                weight_decay=float(g["weight_decay"]))


def fused_config(opt):
    """Hyper-parameters (with "kind": "adam" | "ranger") if the fused engine reproduces `opt` exactly, else None."""
    cfg = adam_config(opt)
    if cfg is not None:
        return dict(cfg, kind="adam")
    if type(opt) is Ranger and len(opt.param_groups) == 1:
        g = opt.param_groups[0]
        return dict(kind="ranger", lr=float(g["lr"]), beta1=float(g["betas"][0]), beta2=float(g["betas"][1]), eps=float(g["eps"]),
                    weight_decay=float(g["weight_decay"]), alpha=float(g["alpha"]), k=int(g["k"]),
                    nsma_threshold=float(g["N_sma_threshhold"]))
    return None


class Ranger(torch.optim.Optimizer):
    """RAdam (variance rectified Adam) + Lookahead(k, alpha); defaults of torch_optimizer.Ranger.  See the module docstring
    for what is and is not verified."""

    def __init__(self, params, lr=1e-3, alpha=0.5, k=6, N_sma_threshhold=5, betas=(0.95, 0.999), eps=1e-5, weight_decay=0):
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1) or not (0 <= alpha <= 1) or k < 1:
            raise ValueError("invalid Ranger hyper-parameter")
        super().__init__(params, dict(lr=lr, alpha=alpha, k=k, N_sma_threshhold=N_sma_threshhold, betas=betas, eps=eps,
                                      weight_decay=weight_decay))

    @staticmethod
    def _init_state(st, p):
        st["step"] = 0
        st["exp_avg"] = torch.zeros_like(p.data)
        st["exp_avg_sq"] = torch.zeros_like(p.data)
        st["slow_buffer"] = p.data.clone()

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        stream = None
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda:
                    raise L.RecnnHipError("recnn_amd.optim.Ranger: parameters must live on the GPU (no CPU fallback)")
                if p.dtype != torch.float32 or not p.data.is_contiguous():
                    raise L.RecnnHipError("recnn_amd.optim.Ranger: parameters must be contiguous float32")
                st = self.state[p]
                if not st:
                    self._init_state(st, p)
                st["step"] = int(st["step"]) + 1
                g = p.grad.data if p.grad.data.is_contiguous() else p.grad.data.contiguous()
                stream = stream or L.current_stream()
                kind, sh = _shadow_of(p)
                L.call("recnn_ranger_flat_shadow", L.ptr(p.data), L.ptr(g), L.ptr(st["exp_avg"]), L.ptr(st["exp_avg_sq"]),
                       L.ptr(st["slow_buffer"]), p.numel(), float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                       float(group["weight_decay"]), float(group["alpha"]), int(group["k"]), float(group["N_sma_threshhold"]),
                       int(st["step"]), 1.0, sh, stream)
                torch.autograd.graph.increment_version(p)
                if kind is not None:
                    _shadow_done(p, kind)
        return loss

    @torch.no_grad()
    def reference_step(self):
        """The same step in plain torch ops (any device): the readable statement of the algorithm the kernel implements;
        tests compare the two."""
        for group in self.param_groups:
            b1, b2 = group["betas"]
            n_max = 2.0 / (1.0 - b2) - 1.0
            for p in group["params"]:
                if p.grad is None:
                    continue
                g = p.grad.data.float()
                st = self.state[p]
                if not st:
                    self._init_state(st, p)
                st["step"] += 1
                t = st["step"]
                m, v = st["exp_avg"], st["exp_avg_sq"]
                v.mul_(b2).addcmul_(g, g, value=1 - b2)
                m.mul_(b1).add_(g, alpha=1 - b1)
                b2t = b2 ** t
                n_sma = n_max - 2 * t * b2t / (1 - b2t)
                if n_sma > group["N_sma_threshhold"]:
                    step_size = math.sqrt((1 - b2t) * (n_sma - 4) / (n_max - 4) * (n_sma - 2) / n_sma * n_max / (n_max - 2)) \
                        / (1 - b1 ** t)
                else:
                    step_size = 1.0 / (1 - b1 ** t)
                if group["weight_decay"] != 0:
                    p.data.add_(p.data, alpha=-group["weight_decay"] * group["lr"])
                if n_sma > group["N_sma_threshhold"]:
                    p.data.addcdiv_(m, v.sqrt().add_(group["eps"]), value=-step_size * group["lr"])
                else:
                    p.data.add_(m, alpha=-step_size * group["lr"])
                if t % group["k"] == 0:
                    slow = st["slow_buffer"]
                    slow.add_(p.data - slow, alpha=group["alpha"])
                    p.data.copy_(slow)
