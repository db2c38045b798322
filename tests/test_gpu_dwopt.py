"""csrc/dwopt.hip: the critic's weight-gradient GEMMs with the optimizer in their epilogue (round 3).

  * against round 2's pipeline (split-batch slabs -> slab-summing Adam, recnn_tune_dw_fuse(0)): the same gradient up to
    the fp32 summation order (whole batch per tile vs 8 slabs), so gradients / moments agree to ~1e-6 and parameters to a
    few per cent of lr -- reference lines: recnn/nn/update/misc.py:42-44 (backward + optimizer step), ddpg.py:95-97.
  * against "gradient arena + apply_kernel" (DWOPT_GRAD through the phase API): BIT FOR BIT -- both apply optim.h's
    contraction-pinned opt_elem to the same fp32 gradient; this is what keeps data-parallel replicas (which must all-reduce
    the arena before stepping) and the single-GPU fused step on the same trajectory.
"""
import pytest
import torch

from oracle import recnn_oracle as O
from tests.helpers import fro_err
from tests.test_gpu_engine import _engine, _init_nets, _rand_batch

pytestmark = pytest.mark.gpu
S, A, H = 1290, 128, 256


def _load(eng, algo, actor, critics, L):
    nets = [(L.NET_POLICY, actor), (L.NET_TARGET_POLICY, actor), (L.NET_VALUE1, critics[0]), (L.NET_TARGET_VALUE1, critics[0])]
    if algo == "td3":
        nets += [(L.NET_VALUE2, critics[1]), (L.NET_TARGET_VALUE2, critics[1])]
    for ni, p in nets:
        eng.load_params(ni, p)
    return [ni for ni, _ in nets]


@pytest.mark.parametrize("algo,B,opt", [("ddpg", 2048, "adam"), ("ddpg", 333, "adam"), ("td3", 777, "adam"), ("ddpg", 2048, "ranger"),
                                        ("ddpg", 5000, "adam")])
def test_fused_dw_optimizer_matches_the_slab_pipeline(cuda, algo, B, opt):
    from recnn_amd import _lib as L
    td3 = algo == "td3"
    actor, critics = _init_nets(3, S, A, H, 2 if td3 else 1)
    batch = _rand_batch(B, S, A, torch.Generator().manual_seed(21))
    kw = dict(lr=1e-3, weight_decay=1e-2)
    if opt == "ranger":
        kw = dict(kind="ranger", lr=1e-3, weight_decay=1e-2, k=2)
    outs = []
    try:
        for fuse in (1, 0):
            L.load().recnn_tune_dw_fuse(fuse)
            eng = _engine(algo, S, A, H, B, "bf16", mask_mode="hash", seed=9)
            nets = _load(eng, algo, actor, critics, L)
            eng.set_hyper(policy_opt=kw, value_opt=kw, policy_every=2, soft_tau=0.05)
            eng.set_counters()
            eng.pack_batch(batch["state"], batch["action"], batch["reward"], batch["next_state"], batch["done"])
            snaps = []
            for t in range(3):                     # steps 0 and 2 are policy steps (soft update inside the epilogue)
                eng.step(B, True, t)
                torch.cuda.synchronize()
                vals = eng.value_nets()
                snaps.append(dict(loss=eng.losses(),
                                  g={ni: eng.grads[ni].clone() for ni in vals},
                                  m={ni: eng.adam_m[ni].clone() for ni in vals},
                                  v={ni: eng.adam_v[ni].clone() for ni in vals},
                                  p={ni: eng.params[ni].clone() for ni in nets}))
            # the compute-type shadows the NEXT forward reads must follow the masters: a learn=False evaluation shows them
            eng.step(B, False, 3)
            snaps.append(dict(loss=eng.losses()))
            outs.append(snaps)
    finally:
        L.load().recnn_tune_dw_fuse(0)
    f, s = outs
    lr = 1e-3
    for t in range(3):
        for k in f[t]["loss"]:
            # (step 0: same weights on both sides; later steps: the two trajectories sit a few sign-flipped Adam moves apart)
            tol = 2e-5 if t == 0 else 1e-3
            assert abs(f[t]["loss"][k] - s[t]["loss"][k]) <= tol * max(abs(s[t]["loss"][k]), 1.0), (t, k, f[t]["loss"], s[t]["loss"])
        for ni in f[t]["g"]:
            if t == 0:   # same weights on both sides: the gradient itself, summation order only
                assert fro_err(f[t]["g"][ni], s[t]["g"][ni]) < 2e-5, (t, ni, fro_err(f[t]["g"][ni], s[t]["g"][ni]))
                assert fro_err(f[t]["m"][ni], s[t]["m"][ni]) < 2e-5
                assert fro_err(f[t]["v"][ni], s[t]["v"][ni]) < 4e-5
            assert torch.isfinite(f[t]["p"][ni]).all()
        for ni in f[t]["p"]:
            d = (f[t]["p"][ni] - s[t]["p"][ni]).abs()
            # Adam moves every element by ~lr per step whatever |g| is: elements whose tiny gradient changes sign between the
            # two summation orders sit 2 lr apart, the rest agree to round-off
            assert d.max().item() <= (2.2 * (t + 1)) * lr + 1e-7, (t, ni, d.max().item())
            assert (d > 0.05 * lr).float().mean().item() < 0.02, (t, ni, (d > 0.05 * lr).float().mean().item())
    for k in f[3]["loss"]:
        assert abs(f[3]["loss"][k] - s[3]["loss"][k]) <= 2e-3 * max(abs(s[3]["loss"][k]), 1.0), (k, f[3]["loss"], s[3]["loss"])


@pytest.mark.parametrize("algo,B", [("ddpg", 2048), ("td3", 1000)])
def test_fused_epilogue_equals_gradient_arena_plus_apply_kernel(cuda, algo, B):
    """recnn_engine_step (optimizer inside the dW epilogue) == phase API (dW writes the arena, apply_kernel steps from it)."""
    from recnn_amd import _lib as L
    td3 = algo == "td3"
    actor, critics = _init_nets(4, S, A, H, 2 if td3 else 1)
    batch = _rand_batch(B, S, A, torch.Generator().manual_seed(22))
    outs = []
    L.load().recnn_tune_dw_fuse(1)
    for mode in ("fused", "phases"):
        eng = _engine(algo, S, A, H, B, "bf16", mask_mode="hash", seed=2)
        nets = _load(eng, algo, actor, critics, L)
        eng.set_hyper(policy_opt=dict(lr=1e-3, weight_decay=1e-2), value_opt=dict(lr=1e-3, weight_decay=1e-2), policy_every=2)
        eng.set_counters()
        eng.pack_batch(batch["state"], batch["action"], batch["reward"], batch["next_state"], batch["done"])
        hist = []
        for t in range(4):
            pol = t % 2 == 0
            if mode == "fused":
                eng.step(B, True, t)
            else:
                eng.value_grads(B, True)
                eng.value_apply(pol, 1.0)
                eng.policy_grads(B, pol)
                if pol:
                    eng.policy_apply(True, 1.0)
                eng.finish(B, True, pol)
            hist.append(eng.losses())
        torch.cuda.synchronize()
        outs.append((hist, {ni: eng.params[ni].clone() for ni in nets}, {ni: eng.adam_v[ni].clone() for ni in eng.value_nets()},
                     {ni: eng.grads[ni].clone() for ni in eng.value_nets()}))
    L.load().recnn_tune_dw_fuse(0)
    assert outs[0][0] == outs[1][0], (outs[0][0], outs[1][0])
    for part in (1, 2, 3):
        for ni in outs[0][part]:
            assert torch.equal(outs[0][part][ni], outs[1][part][ni]), (part, ni, (outs[0][part][ni] - outs[1][part][ni]).abs().max().item())
