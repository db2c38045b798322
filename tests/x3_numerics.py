"""Numerics probe (CPU, not collected by pytest): how far does the DDPG loss curve move when every contraction of the
step is evaluated in a reduced operand format instead of fp32?

Formats: "bf16" (operands rounded to bf16, fp32 accumulation: the MFMA-bf16 path), "x3" (split bf16: x = hi + lo with
hi = bf16(x), lo = bf16(x - hi); products hi*hi + hi*lo + lo*hi, fp32 accumulation), "x2a" / "x2w" (two products: only the first / only the second operand
split), "x4" (+ lo*lo), "x6" (three-way split); "a/b/c" = forward / dX / dW formats.  The oracle's own arithmetic (fp32) is the reference curve.

    python tests/x3_numerics.py [steps] [rows]
"""
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import recnn_oracle as O  # noqa: E402


def split(x):
    hi = x.bfloat16().float()
    lo = (x - hi).bfloat16().float()
    return hi, lo


def mm_fmt(fmt):
    def mm(a, b):
        if fmt == "fp32":
            return a @ b
        if fmt == "bf16":
            return a.bfloat16().float() @ b.bfloat16().float()
        if fmt == "fp32r":   # fp32 with another summation order (k halves added at the end): the fp32 noise floor
            h = a.shape[1] // 2
            return a[:, :h] @ b[:h] + a[:, h:] @ b[h:]
        if fmt == "x6":      # three-way split, products down to 2^-24
            a0 = a.bfloat16().float(); a1 = (a - a0).bfloat16().float(); a2 = (a - a0 - a1).bfloat16().float()
            b0 = b.bfloat16().float(); b1 = (b - b0).bfloat16().float(); b2 = (b - b0 - b1).bfloat16().float()
            return ((a2 @ b0 + a0 @ b2) + a1 @ b1) + (a1 @ b0 + a0 @ b1) + a0 @ b0
        ah, al = split(a)
        bh, bl = split(b)
        if fmt == "x3":
            return (al @ bh + ah @ bl) + ah @ bh
        if fmt == "x2a":     # two MFMAs per product: activations split, weights rounded to bf16
            return al @ bh + ah @ bh
        if fmt == "x2w":     # two MFMAs per product: weights split, activations rounded to bf16
            return ah @ bl + ah @ bh
        if fmt == "x4":
            return (al @ bl + al @ bh + ah @ bl) + ah @ bh
        raise ValueError(fmt)
    return mm


def patched(fmt):
    # "a/b/c": forward / dX / dW formats
    parts = fmt.split("/")
    mm = mm_fmt(parts[0])
    mmx = mm_fmt(parts[1] if len(parts) > 1 else parts[0])
    mmw = mm_fmt(parts[2] if len(parts) > 2 else parts[0])

    def mlp_forward(p, x, m1=None, m2=None):
        h1 = O._drop(torch.relu(mm(x, p["w1"].t()) + p["b1"]), m1)
        h2 = O._drop(torch.relu(mm(h1, p["w2"].t()) + p["b2"]), m2)
        out = mm(h2, p["w3"].t()) + p["b3"]
        return out, (x, h1, h2)

    def mlp_backward(p, cache, dout, train=True, need_dx=False, need_dw=True):
        x, h1, h2 = cache
        s = 2.0 if train else 1.0
        g = {}
        if need_dw:
            g["w3"] = mmw(dout.t(), h2)
            g["b3"] = dout.sum(0)
        dz2 = mmx(dout, p["w3"]) * ((h2 > 0).to(dout.dtype) * s)
        if need_dw:
            g["w2"] = mmw(dz2.t(), h1)
            g["b2"] = dz2.sum(0)
        dz1 = mmx(dz2, p["w2"]) * ((h1 > 0).to(dout.dtype) * s)
        if need_dw:
            g["w1"] = mmw(dz1.t(), x)
            g["b1"] = dz1.sum(0)
        dx = mmx(dz1, p["w1"]) if need_dx else None
        return (g if need_dw else None), dx, {"dz2": dz2, "dz1": dz1}
    return mlp_forward, mlp_backward


def make_nets(seed):
    g = torch.Generator().manual_seed(seed)

    def lin(o, i, w=None):
        k = 1.0 / i ** 0.5 if w is None else w
        return (torch.rand(o, i, generator=g) * 2 - 1) * k, (torch.rand(o, generator=g) * 2 - 1) * k
    def net(i, o, w):
        w1, b1 = lin(256, i)
        w2, b2 = lin(256, 256)
        w3, b3 = lin(o, 256, w)
        return {"w1": w1, "b1": b1, "w2": w2, "b2": b2, "w3": w3, "b3": b3}
    return net(1290, 128, 6e-1), net(1418, 1, 54e-2)


def batches(n, rows, seed):
    g = torch.Generator().manual_seed(seed)
    out = []
    for _ in range(n):
        def st():
            e = torch.randn(rows, 1280, generator=g)
            r = (torch.randint(1, 11, (rows, 10), generator=g).float() * 0.5 - 2.5) * 2
            return torch.cat([e, r], 1)
        b = {"state": st(), "next_state": st(), "action": torch.randn(rows, 128, generator=g),
             "reward": (torch.randint(1, 11, (rows,), generator=g).float() * 0.5 - 2.5) * 2,
             "done": (torch.rand(rows, generator=g) < 0.03).float()}
        masks = [(torch.rand(rows, 256, generator=g) < 0.5).to(torch.uint8) for _ in range(6)]
        out.append((b, masks))
    return out


def run(fmt, data):
    fwd, bwd = patched(fmt)
    O.mlp_forward, O.mlp_backward = fwd, bwd
    pol, val = make_nets(5)
    st = O.DDPGState.create(pol, val, O.AdamState(lr=1e-5, weight_decay=1e-2), O.AdamState(lr=1e-5, weight_decay=1e-2))
    hist = [O.ddpg_step(st, b, m, step=i, learn=True) for i, (b, m) in enumerate(data)]
    return hist, st


if __name__ == "__main__":
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rows = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
    fmts = sys.argv[3].split(",") if len(sys.argv) > 3 else ["bf16", "x3", "x4"]
    torch.set_num_threads(8)
    data = batches(steps, rows, 7)
    ref, rst = run("fp32", data)
    for fmt in fmts:
        got, gst = run(fmt, data)
        bad = tot = exc = 0
        mx = fro = 0.0
        for net in ("policy", "value"):
            opt = getattr(rst, net + "_opt")
            for k in O.PARAM_ORDER:
                a, b = getattr(gst, net)[k], getattr(rst, net)[k]
                vhat = (opt.v[k] / (1.0 - opt.beta2 ** opt.t)).sqrt()
                epsr = vhat < 1e3 * opt.eps
                dev = (a - b).abs()
                bad += int(((dev > 1e-4 * b.abs() + 1e-4 * b.pow(2).mean().sqrt()) & ~epsr).sum())
                exc += int(epsr.sum())
                tot += b.numel()
                mx = max(mx, float(dev.max()))
                fro = max(fro, float((a - b).norm() / b.norm()))
        print(f"{fmt:14s} audit: outside {bad}/{tot} = {bad / tot:.4f} (<=0.01)  excluded {exc / tot:.4f}  max_dev {mx / 1e-5:.2f} lr (<=20)  fro {fro:.2e} (<=1e-4)")
        for k in ("value", "policy"):
            devs = [abs(a[k] - b[k]) / (abs(b[k]) + 1e-6) for a, b in zip(got, ref)]
            print(f"{fmt:14s} {k:6s} worst {max(devs):.3e}  at step {devs.index(max(devs))}  last {devs[-1]:.3e}")
