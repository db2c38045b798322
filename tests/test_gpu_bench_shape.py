"""GPU parity of the BENCHMARKED configuration itself (bench.py: rows 2048, 256 users per batch, policy_step 10, hash
dropout masks, hipGraph replay through the run-graph family with look-ahead gather and deferred policy-loss forward).

  * `Algo.run(n)` in one call == the same steps replayed as several `run()` calls that start mid-cycle (5 + 20 + rest:
    exactly what `bench.py --warmup 5 --steps 20` does) == the reference-shaped loop `update(batch); step()` on the same
    batches -- parameters bit for bit, per-step losses to summation order (bf16 AND fp32).
  * fp32: every one of 200 steps' losses within 1e-4 of the CPU oracle driven with the same batches and the dumped
    hash masks (the loss curve of north_star), final parameters element-wise at rtol 1e-4 with the Adam eps-regime
    elements excluded and counted.
  * bf16: the deviation of the same loss curve from the fp32 oracle is measured and bounded (reported, not 1e-4).
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import recnn_oracle as O
from tests.helpers import within

pytestmark = pytest.mark.gpu

ROWS, UPB, PE = 2048, 256, 10
SEED = 4242
# bf16 loss curve against the fp32 oracle over 65 steps: ceilings; the asserted bounds are tests.helpers.BF16_BOUNDS (2.5x measured)
BF16_CURVE_VALUE = 8e-4
BF16_CURVE_POLICY = 3e-3
FLIP_LR = 3.5           # TD3 audit: deviations of at most this many lr are Adam sign flips (counted separately)
X3_AUDIT_MAX = 0.01     # bf16x3: fraction of parameter elements outside rtol 1e-4 after 200 steps: 0.0042 measured (fp32: 0.0058), same bound


def _bench_env(recnn_amd, cuda, n_users, seed=3, n_items=3000):
    rng = np.random.default_rng(seed)
    lens = rng.integers(20, 61, size=n_users).astype(np.int64)          # every user has >= 10 windows
    off = np.zeros(n_users + 1, dtype=np.int64)
    off[1:] = np.cumsum(lens)
    total = int(off[-1])
    items = rng.integers(0, n_items, size=total, dtype=np.int32)
    ratings = (2.0 * (rng.integers(1, 11, size=total) * 0.5 - 2.5)).astype(np.float32)
    table = torch.randn(n_items, 128, generator=torch.Generator().manual_seed(seed))
    env = recnn_amd.data.env.FrameEnv.from_store(table, items, ratings, off, frame_size=10, batch_size=25, device=cuda,
                                                 test_fraction=0.0, rows_per_batch=ROWS)
    return env, table


def _make_algo(recnn_amd, cuda, env, dtype):
    from recnn_amd.nn import fused
    fused.set_defaults(dtype=dtype, mask_mode="hash", seed=SEED)
    torch.manual_seed(12)
    ddpg = recnn_amd.nn.DDPG(recnn_amd.nn.Actor(1290, 128, 256, 6e-1), recnn_amd.nn.Critic(1290, 128, 256, 54e-2)).to(cuda)
    assert ddpg.params["policy_step"] == PE
    torch.manual_seed(99)                                  # the epoch permutation comes from the CPU generator
    ddpg.attach_env(env, rows_per_batch=ROWS, users_per_batch=UPB)
    return ddpg


def _snapshot(ddpg):
    return {n: {k: v.detach().clone() for k, v in ddpg.nets[n].state_dict().items()}
            for n in ("policy_net", "value_net", "target_policy_net", "target_value_net")}


def _hash_masks(L, step, cuda):
    out = []
    for stream in range(6):
        m = torch.zeros(ROWS, 256, dtype=torch.uint8, device=cuda)
        L.call("recnn_hash_mask_dump", SEED, int(step), stream, ROWS, 256, L.ptr(m), L.current_stream())
        out.append(m)
    torch.cuda.synchronize()
    return [m.cpu() for m in out]


@pytest.mark.parametrize("dtype,n", [("bf16", 65), ("fp32", 200), ("bf16x3", 200)])
def test_bench_shape_run_equals_loop_and_oracle(cuda, dtype, n):
    import recnn_amd
    from recnn_amd import _lib as L
    # bf16 (the benchmarked dtype) on a table of ML20M's size (26,744 items: bench.py's), fp32 on a 3,000-item one
    env, table = _bench_env(recnn_amd, cuda, n_users=(n + 2) * UPB, n_items=26744 if dtype == "bf16" else 3000)
    results = {}
    for mode in ("one_call", "pieces", "prepared", "loop"):
        ddpg = _make_algo(recnn_amd, cuda, env, dtype)
        ctx = ddpg._fused_ctx
        assert ctx.sampler["n_batches"] >= n and ctx.engine.dtype == dtype
        if mode == "one_call":
            out, hist = ddpg.run(n, history=True)
        elif mode in ("pieces", "prepared"):
            hist = []
            if mode == "prepared":                         # bench.py: the timed 20-step call is ONE made-to-order run graph
                ddpg.prepare_run(20, first_step=5)
                assert ctx._run_seen[(5, 20)] == 2
            for k in (5, 20, n - 25):                      # the driver's bench: warm-up 5, then 20 steps from step 5
                out, h = ddpg.run(k, history=True)
                hist += h
        else:
            perm = ctx.perm.cpu().numpy()
            ost = O.DDPGState.create(O.params_from_module(ddpg.nets["policy_net"]), O.params_from_module(ddpg.nets["value_net"]),
                                     O.AdamState(lr=1e-5, weight_decay=1e-2), O.AdamState(lr=1e-5, weight_decay=1e-2))
            hist, ohist = [], []
            for i in range(n):
                batch = env.collate_users([int(u) for u in perm[i * UPB:(i + 1) * UPB]])
                assert batch["state"].shape[0] == ROWS
                out = ddpg.update(batch, learn=True)
                hist.append(dict(out))
                ohist.append(O.ddpg_step(ost, {k: batch[k].float().cpu() for k in ("state", "action", "reward", "next_state", "done")},
                                         _hash_masks(L, i, cuda), step=i, learn=True))
                ddpg.step()
            results["oracle"] = (ohist, ost)
        torch.cuda.synchronize()
        assert ddpg._step == n and out["step"] == n - 1 and len(hist) == n
        assert ctx.engine.counters()[0] == n
        results[mode] = (hist, _snapshot(ddpg))
    # ---- the three ways of running the same n steps: parameters bit for bit
    for other in ("pieces", "prepared", "loop"):
        for net, sd in results["one_call"][1].items():
            for k, v in sd.items():
                assert torch.equal(v, results[other][1][net][k]), (other, net, k)
        for a, b in zip(results["one_call"][0], results[other][0]):
            assert a["step"] == b["step"]
            for k in ("value", "policy"):
                assert abs(a[k] - b[k]) <= 1e-5 * max(abs(b[k]), 1.0), (other, a, b)
    # ---- against the CPU oracle (fp32 reference arithmetic, same batches, same masks)
    ohist, ost = results["oracle"]
    worst = {"value": 0.0, "policy": 0.0}
    for got, ref in zip(results["loop"][0], ohist):
        for k in worst:
            worst[k] = max(worst[k], abs(got[k] - ref[k]) / (abs(ref[k]) + 1e-6))
    report = {"dtype": dtype, "steps": n, "worst_rel_loss_dev": worst}
    if dtype != "bf16":          # fp32 and split bf16 ("bf16x3"): north_star's 1e-4 on every step of the loss curve
        assert worst["value"] <= 1e-4 and worst["policy"] <= 1e-4, worst
        # final parameters, element-wise: |got - ref| <= 1e-4 |ref| + 1e-4 rms(ref).  Elements whose gradient scale
        # sqrt(v_hat) sits in Adam's eps regime (< 1e3 eps: the update lr*m/(sqrt(v)+eps) amplifies round-off by up to
        # lr/eps there) are excluded -- and counted.  Of the rest a small fraction still misses the bound, inherently:
        # Adam's first steps move every element by ~lr*sign(g); a relu gate whose pre-activation sits within fp32
        # round-off of zero (a few per step; a thread-count change flips them in the reference too) shifts one hidden
        # unit's ~1300 weight gradients by one row's contribution, which flips the sign of those that are smaller than
        # it -- each such element then sits 2 lr away for good.  Measured: ~0.6 % of 858k elements after 200 steps, all
        # within a few lr.  So: <= 1 % outside the bound, nothing further than 20 lr, Frobenius error <= 1e-4.
        excluded = failed = total = 0
        max_dev = fro = 0.0
        for net, refp, opt in (("policy_net", ost.policy, ost.policy_opt), ("value_net", ost.value, ost.value_opt)):
            sd = results["loop"][1][net]
            for k, name in zip(O.PARAM_ORDER, ("linear1.weight", "linear1.bias", "linear2.weight", "linear2.bias",
                                               "linear3.weight", "linear3.bias")):
                got, ref = sd[name].float().cpu(), refp[k]
                vhat = (opt.v[k] / (1.0 - opt.beta2 ** opt.t)).sqrt()
                eps_regime = vhat < 1e3 * opt.eps
                dev = (got - ref).abs()
                bad = dev > 1e-4 * ref.abs() + 1e-4 * ref.pow(2).mean().sqrt()
                excluded += int(eps_regime.sum())
                failed += int((bad & ~eps_regime).sum())
                total += ref.numel()
                max_dev = max(max_dev, float(dev.max()))
                fro = max(fro, float((got - ref).norm() / ref.norm()))
        report.update(param_elements=total, eps_regime_excluded=excluded, outside_rtol_1e4=failed, max_abs_dev=max_dev,
                      max_abs_dev_in_lr=max_dev / 1e-5, worst_frobenius=fro)
        # (split bf16 carries 16-17 significand bits per operand: more Adam sign flips than fp32's 24, same mechanism; measured
        # on MI355X: see X3_AUDIT_MAX)
        assert failed <= (0.01 if dtype == "fp32" else X3_AUDIT_MAX) * total, report
        assert excluded <= 0.05 * total, report
        assert max_dev <= 20 * 1e-5, report
        assert fro <= 1e-4, report
        for net, refp in (("target_policy_net", ost.target_policy), ("target_value_net", ost.target_value)):
            sd = results["loop"][1][net]
            for k, name in zip(O.PARAM_ORDER, ("linear1.weight", "linear1.bias", "linear2.weight", "linear2.bias",
                                               "linear3.weight", "linear3.bias")):
                got, ref = sd[name].float().cpu(), refp[k]
                assert float((got - ref).norm() / ref.norm()) <= 1e-5, (net, k)      # tau = 1e-3 times the deviations above
    else:
        # bf16 compute (fp32 master weights / accumulation): measured, bounded, NOT claimed as 1e-4
        within("bench_shape/bf16/loss_curve_value", worst["value"], BF16_CURVE_VALUE)
        within("bench_shape/bf16/loss_curve_policy", worst["policy"], BF16_CURVE_POLICY)
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, f"bench_shape_parity_{dtype}.json"), "w") as f:
        json.dump(report, f)
    print("bench-shape parity:", json.dumps(report))


def test_short_segment_frozen_launches_on_64_row_workgroups(cuda):
    """tuning.frozen_half (round 6): a request that starts and ends inside policy cycles -- the driver's 5 + 20 steps are cycle segments of
    6 + 10 + 4 -- runs the frozen networks of its short segments on 64-row workgroups (csrc/mlpf.hip, wave tile 32 x 32) instead of
    128-row ones: every parameter and the loss history bit for bit, for several request lengths (segments of 1..10 steps)."""
    import recnn_amd
    from recnn_amd._tune import set_default_tuning
    env, _ = _bench_env(recnn_amd, cuda, n_users=80 * UPB, n_items=3000)

    def run(half):
        set_default_tuning(frozen_half=half)
        try:
            ddpg = _make_algo(recnn_amd, cuda, env, "bf16")
            assert ddpg._fused_ctx.engine.tuning.frozen_half == half
            hist = []
            first = 0
            for k in (5, 20, 23, 27):                      # 20 from step 5: 6 + 10 + 4; 23 from 25: 6 + 10 + 7; 27 from 48: 3 + 10 + 10 + 4
                if k >= 20:
                    ddpg.prepare_run(k, first_step=first)
                _, h = ddpg.run(k, history=True)
                hist += h
                first += k
            torch.cuda.synchronize()
            return hist, _snapshot(ddpg)
        finally:
            set_default_tuning(frozen_half=None)
    (h0, p0), (h1, p1) = run(0), run(1)
    for net, sd in p0.items():
        for k, v in sd.items():
            assert torch.equal(v, p1[net][k]), (net, k)
    assert h0 == h1


def test_run_interleaved_with_test_updates(cuda):
    """`algo.update(env.test_batch(), learn=False)` between two `run()` calls evaluates the batch it was given (not a
    sampler batch), leaves the sampler cursor and the optimizer counters alone, and the next `run()` picks up where the
    previous one stopped (engine.hip stage_batch / graph_rows after bind_batch)."""
    import recnn_amd
    from recnn_amd.nn import fused
    env, table = _bench_env(recnn_amd, cuda, n_users=40 * UPB)
    fused.set_defaults(dtype="bf16", mask_mode="hash", seed=SEED)
    torch.manual_seed(12)
    ddpg = recnn_amd.nn.DDPG(recnn_amd.nn.Actor(1290, 128, 256, 6e-1), recnn_amd.nn.Critic(1290, 128, 256, 54e-2)).to(cuda)
    torch.manual_seed(99)
    ddpg.attach_env(env, rows_per_batch=ROWS, users_per_batch=UPB)
    ctx = ddpg._fused_ctx
    ddpg.run(13)
    c0 = ctx.engine.counters()
    assert c0 == (13, 2, 13, 0) and int(ctx.engine.cursor.item()) == 13
    w0 = ddpg.nets["value_net"].linear1.weight.detach().clone()
    tb = env.collate_users(list(range(7, 7 + 40)))                  # a packed FrameEnv batch: bound in place
    tb2 = env.collate_users(list(range(200, 200 + 40)))
    l1 = ddpg.update(tb, learn=False)
    l2 = ddpg.update(tb2, learn=False)
    assert np.isfinite(l1["value"]) and np.isfinite(l2["value"]) and l1["value"] != l2["value"]
    c1 = ctx.engine.counters()
    assert c1[1:] == c0[1:] and c1[0] == 15                          # two evaluations, no optimizer step
    assert int(ctx.engine.cursor.item()) == 13 and ctx.sampler["cursor"] == 13
    assert torch.equal(w0, ddpg.nets["value_net"].linear1.weight)
    out = ddpg.run(12)                                               # graphs are rebuilt on the engine's own rows
    torch.cuda.synchronize()
    assert np.isfinite(out["value"]) and out["step"] == 24
    assert ctx.engine.counters() == (27, 3, 25, 0) and int(ctx.engine.cursor.item()) == 25   # policy steps 0, 10, 20
    assert not torch.equal(w0, ddpg.nets["value_net"].linear1.weight)


def test_learn_false_update_sees_its_own_batch(cuda):
    """With a sampler attached, update(batch, learn=False) on two DIFFERENT batches gives two different losses, equal to
    what an engine without a sampler reports for them."""
    import recnn_amd
    from recnn_amd.nn import fused
    env, table = _bench_env(recnn_amd, cuda, n_users=8 * UPB)
    fused.set_defaults(dtype="fp32", mask_mode="none", seed=SEED)
    torch.manual_seed(12)
    ddpg = recnn_amd.nn.DDPG(recnn_amd.nn.Actor(1290, 128, 256, 6e-1), recnn_amd.nn.Critic(1290, 128, 256, 54e-2)).to(cuda)
    for m in (ddpg.nets["policy_net"], ddpg.nets["value_net"]):
        m.eval()                                                    # no dropout: the evaluation is deterministic
    b1 = env.collate_users(list(range(0, 60)))
    b2 = env.collate_users(list(range(100, 160)))
    ref1, ref2 = ddpg.update(b1, learn=False), ddpg.update(b2, learn=False)
    assert ref1["value"] != ref2["value"]
    torch.manual_seed(99)
    ddpg.attach_env(env, rows_per_batch=ROWS, users_per_batch=UPB)
    got1, got2 = ddpg.update(b1, learn=False), ddpg.update(b2, learn=False)
    assert got1["value"] == ref1["value"] and got2["value"] == ref2["value"], (got1, ref1, got2, ref2)
    assert got1["policy"] == ref1["policy"] and got2["policy"] == ref2["policy"]


def test_td3_b4096_run_graphs_equal_pieces_and_update_loop(cuda):
    """BASELINE configs[2] (TD3, twin critics, delayed actor, 4096 rows per step, bf16, hash dropout masks and on-device target
    noise): `Algo.run(45)` in one call == the same steps as run(5); run(20); run(20) (mid-cycle starts, made-to-order graph) ==
    the reference-shaped loop `update(batch); step()` on the same batches -- parameters of all six networks bit for bit, per-step
    losses (value1, value2, policy) to summation order."""
    import recnn_amd
    from recnn_amd.nn import fused
    rows, upb, n = 4096, 512, 65          # run(65) = a 60-step multi-cycle graph (cycle mode: frozen networks hoisted) + 5 steps
    env, _ = _bench_env(recnn_amd, cuda, n_users=(n + 2) * upb, seed=5)
    env.rows_per_batch = rows                  # (the helper builds a 2048-row env)
    names = ("policy_net", "value_net1", "value_net2", "target_policy_net", "target_value_net1", "target_value_net2")
    results = {}
    for mode in ("one_call", "pieces", "loop"):
        fused.set_defaults(dtype="bf16", mask_mode="hash", seed=SEED)
        torch.manual_seed(21)
        td3 = recnn_amd.nn.TD3(recnn_amd.nn.Actor(1290, 128, 256, 6e-1), recnn_amd.nn.Critic(1290, 128, 256, 54e-2),
                               recnn_amd.nn.Critic(1290, 128, 256, 54e-2)).to(cuda)
        every = td3.params["policy_update"]
        torch.manual_seed(77)
        td3.attach_env(env, rows_per_batch=rows, users_per_batch=upb)
        ctx = td3._fused_ctx
        assert ctx.sampler["n_batches"] >= n
        if mode == "one_call":
            out, hist = td3.run(n, history=True)
        elif mode == "pieces":
            hist = []
            td3.prepare_run(20, first_step=5)
            for k in (5, 20, n - 25):
                out, h = td3.run(k, history=True)
                hist += h
        else:
            perm = ctx.perm.cpu().numpy()
            hist = []
            for i in range(n):
                batch = env.collate_users([int(u) for u in perm[i * upb:(i + 1) * upb]])
                assert batch["state"].shape[0] == rows
                out = td3.update(batch, learn=True)
                hist.append(dict(out))
                td3.step()
        torch.cuda.synchronize()
        assert td3._step == n and len(hist) == n
        results[mode] = (hist, {nm: {k: v.detach().clone() for k, v in td3.nets[nm].state_dict().items()} for nm in names}, every)
    for other in ("pieces", "loop"):
        for net, sd in results["one_call"][1].items():
            for k, v in sd.items():
                assert torch.equal(v, results[other][1][net][k]), (other, net, k)
        for a, b in zip(results["one_call"][0], results[other][0]):
            assert a["step"] == b["step"]
            for k in ("value1", "value2", "policy"):
                assert abs(a[k] - b[k]) <= 1e-5 * max(abs(b[k]), 1.0), (other, k, a, b)
    assert all(np.isfinite(h[k]) for h in results["one_call"][0] for k in ("value1", "value2", "policy"))


@pytest.mark.parametrize("dtype", ["fp32", "bf16x3"])
def test_td3_b4096_loss_curve_and_parameters_vs_oracle(cuda, dtype):
    """BASELINE configs[2] against the CPU oracle (it was pinned for 2 steps only): TD3, 4096 rows, fp32 and split bf16, 200
    steps of the reference-shaped loop `update(batch); step()` with hash dropout masks and on-device target noise, the oracle
    (recnn/nn/update/td3.py:66-150 restated, pinned to the real reference by oracle/make_golden.py) driven with the same
    batches, the dumped masks and the dumped noise draw of every step.  Every step's three losses within north_star's 1e-4;
    final parameters of all six networks element-wise at rtol 1e-4 (Adam eps-regime elements excluded and counted, as in the
    DDPG test above), nothing further than 20 lr, Frobenius <= 1e-4."""
    import recnn_amd
    from recnn_amd import _lib as L
    from recnn_amd.nn import fused
    rows, upb, n = 4096, 512, 200
    env, _ = _bench_env(recnn_amd, cuda, n_users=(n + 2) * upb, seed=7)
    env.rows_per_batch = rows
    fused.set_defaults(dtype=dtype, mask_mode="hash", seed=SEED)
    torch.manual_seed(23)
    td3 = recnn_amd.nn.TD3(recnn_amd.nn.Actor(1290, 128, 256, 6e-1), recnn_amd.nn.Critic(1290, 128, 256, 54e-2),
                           recnn_amd.nn.Critic(1290, 128, 256, 54e-2)).to(cuda)
    torch.manual_seed(78)
    td3.attach_env(env, rows_per_batch=rows, users_per_batch=upb)
    ctx = td3._fused_ctx
    eng = ctx.engine
    assert eng.dtype == dtype and ctx.sampler["n_batches"] >= n
    lr, wd = 1e-5, 1e-2
    for k in ("policy_optimizer", "value_optimizer1", "value_optimizer2"):
        g = td3.optimizers[k].param_groups[0]
        assert abs(g["lr"] - lr) < 1e-12 and abs(g["weight_decay"] - wd) < 1e-12
    ost = O.TD3State.create(O.params_from_module(td3.nets["policy_net"]), O.params_from_module(td3.nets["value_net1"]),
                            O.params_from_module(td3.nets["value_net2"]), O.AdamState(lr=lr, weight_decay=wd),
                            O.AdamState(lr=lr, weight_decay=wd), O.AdamState(lr=lr, weight_decay=wd))
    for k in ("gamma", "noise_std", "noise_clip", "soft_tau", "policy_update"):
        ost.params[k] = td3.params[k]

    def masks_of(step):
        out = []
        for stream in range(8):
            m = torch.zeros(rows, 256, dtype=torch.uint8, device=cuda)
            L.call("recnn_hash_mask_dump", SEED, int(step), stream, rows, 256, L.ptr(m), L.current_stream())
            out.append(m)
        torch.cuda.synchronize()
        return [m.cpu() for m in out]

    perm = ctx.perm.cpu().numpy()
    worst = {"value1": 0.0, "value2": 0.0, "policy": 0.0}
    for i in range(n):
        batch = env.collate_users([int(u) for u in perm[i * upb:(i + 1) * upb]])
        assert batch["state"].shape[0] == rows
        got = td3.update(batch, learn=True)
        noise = eng.buffer("noise", rows).cpu()                # this step's unclipped N(0, noise_std) draw (on-device generator)
        ref = O.td3_step(ost, {k: batch[k].float().cpu() for k in ("state", "action", "reward", "next_state", "done")}, noise,
                         masks_of(i), step=i, learn=True)
        for k in worst:
            worst[k] = max(worst[k], abs(got[k] - ref[k]) / (abs(ref[k]) + 1e-6))
        td3.step()
    assert float(noise.std()) > 0.3 and float(noise.abs().max()) < 6.0         # a real Gaussian draw went through the oracle
    report = {"algo": "td3", "rows": rows, "dtype": dtype, "steps": n, "worst_rel_loss_dev": worst}
    for k in worst:
        assert worst[k] <= 1e-4, worst
    excluded = failed = flips = total = 0
    max_dev = fro = 0.0
    names = ("linear1.weight", "linear1.bias", "linear2.weight", "linear2.bias", "linear3.weight", "linear3.bias")
    for net, refp, opt in (("policy_net", ost.policy, ost.policy_opt), ("value_net1", ost.value1, ost.value_opt1),
                           ("value_net2", ost.value2, ost.value_opt2)):
        sd = td3.nets[net].state_dict()
        for k, name in zip(O.PARAM_ORDER, names):
            gotp, refk = sd[name].float().cpu(), refp[k]
            vhat = (opt.v[k] / (1.0 - opt.beta2 ** opt.t)).sqrt()
            eps_regime = vhat < 1e3 * opt.eps
            dev = (gotp - refk).abs()
            bad = dev > 1e-4 * refk.abs() + 1e-4 * refk.pow(2).mean().sqrt()
            excluded += int(eps_regime.sum())
            # An Adam step moves an element by ~ lr * sign(g) whatever |g| is: an element whose gradient changes sign between two
            # equally exact summation orders at ONE of the 200 steps sits 2 lr away for good.  Those are counted on their own
            # (deviation of at most FLIP_LR Adam quanta); everything else that misses rtol 1e-4 is `failed`.
            flip = bad & ~eps_regime & (dev <= FLIP_LR * lr)
            flips += int(flip.sum())
            failed += int((bad & ~eps_regime & ~flip).sum())
            total += refk.numel()
            max_dev = max(max_dev, float(dev.max()))
            fro = max(fro, float((gotp - refk).norm() / refk.norm()))
    report.update(param_elements=total, eps_regime_excluded=excluded, adam_sign_flips=flips, outside_rtol_1e4_other=failed,
                  outside_rtol_1e4=failed + flips, max_abs_dev=max_dev, max_abs_dev_in_lr=max_dev / lr, worst_frobenius=fro)
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, f"bench_shape_parity_td3_{dtype}.json"), "w") as f:
        json.dump(report, f)
    print("TD3 bench-shape parity:", json.dumps(report))
    # Elements outside rtol 1e-4 (eps regime excluded): fp32 0.27 %, split bf16 1.18 % (measured, round 4).  VERDICT r4: the fp32 test's
    # 1 % bound stays for BOTH types on everything that is not an Adam sign flip (measured: none -- every deviation is within 3.0 lr);
    # the flips are counted on their own: at most 1 % in fp32 (as before), 2 % in split bf16, whose pre-activations carry ~1e-5
    # instead of ~1e-7 of relative error and whose two critics train on min(Q1', Q2'), an argmin that flips with them.
    # (in fp32 the two buckets share ONE 1 % budget -- a regression that moves many elements by < 3.5 lr must not pass on the
    #  flip allowance; the separate 2 % flip budget is split bf16's only.)
    assert failed <= 0.01 * total, report
    if dtype == "fp32":
        assert failed + flips <= 0.01 * total, report
    else:
        assert flips <= 0.02 * total, report
    assert excluded <= 0.05 * total, report
    assert max_dev <= 20 * lr, report
    assert fro <= 1e-4, report
    for net, refp in (("target_policy_net", ost.target_policy), ("target_value_net1", ost.target_value1),
                      ("target_value_net2", ost.target_value2)):
        sd = td3.nets[net].state_dict()
        for k, name in zip(O.PARAM_ORDER, names):
            gotp, refk = sd[name].float().cpu(), refp[k]
            assert float((gotp - refk).norm() / refk.norm()) <= 1e-5, (net, k)


def test_run_on_a_caller_owned_stream_equals_run_from_the_default_stream(cuda):
    """Algo.run launches its graphs on the caller's stream when that is a real stream (bench.py) and hops to a private stream
    from the default one: same steps, parameters bit for bit, and the caller's stream order is respected (a tensor written on
    that stream before run() and read after it)."""
    import recnn_amd
    env, _ = _bench_env(recnn_amd, cuda, n_users=40 * UPB)
    snaps = []
    for owned in (False, True):
        ddpg = _make_algo(recnn_amd, cuda, env, "bf16")
        if owned:
            st = torch.cuda.Stream(device=cuda)
            with torch.cuda.stream(st):
                ddpg.run(7)
                out = ddpg.run(25)
            st.synchronize()
        else:
            ddpg.run(7)
            out = ddpg.run(25)
        torch.cuda.synchronize()
        assert out["step"] == 31
        snaps.append(_snapshot(ddpg))
    for net, sd in snaps[0].items():
        for k, v in sd.items():
            assert torch.equal(v, snaps[1][net][k]), (net, k)


def test_reference_shaped_loop_on_planned_batches_equals_run(cuda):
    """`for batch in algo.batches(n): loss = algo.update(batch); algo.step()` (the reference's loop: update returns the losses
    dict, here with lazy values, steps are queued and replayed 60 at a time) == `algo.run(n)`: all four networks bit for bit,
    every step's losses equal to run's history; reading a loss mid-way flushes the queue; handles out of order are refused."""
    import recnn_amd
    n = 147
    env, _ = _bench_env(recnn_amd, cuda, n_users=UPB * 40)
    a = _make_algo(recnn_amd, cuda, env, "bf16")
    _, hist = a.run(n, history=True)
    want = _snapshot(a)
    b = _make_algo(recnn_amd, cuda, env, "bf16")
    got = []
    for i, batch in enumerate(b.batches(n)):
        loss = b.update(batch, learn=True)
        b.step()
        got.append(loss)
        if i == 70:
            assert float(loss["value"]) == hist[70]["value"]           # resolves now: flushes the 11 queued steps
    b.flush()
    torch.cuda.synchronize()
    snap = _snapshot(b)
    for net in want:
        for k in want[net]:
            assert torch.equal(want[net][k], snap[net][k]), (net, k)
    assert b._step == a._step == n
    for i, (l, h) in enumerate(zip(got, hist)):
        # (the policy loss of a step is a sum over rows whose grouping depends on the step's place in its run graph -- per-row
        # Q of a deferred forward vs per-workgroup partial dots --: equal to fp32 summation order, like the reference's .mean())
        assert l["step"] == i and float(l["value"]) == h["value"] and abs(float(l["policy"]) - h["policy"]) <= 1e-6 * abs(h["policy"]), (i, h)
    assert f"{got[3]['value']:.4f}" == f"{hist[3]['value']:.4f}" and got[5]["policy"] + 1.0 == hist[5]["policy"] + 1.0
    it = b.batches()
    first = next(it)
    b.update(first); b.step()
    with pytest.raises(RuntimeError, match="out of order"):
        b.update(first)                                    # a handle is good for one update
    assert first["state"].shape == (ROWS, 1290) and first["reward"].shape == (ROWS,)     # a handle materialises on demand


def test_train_dataloader_driven_by_the_algo_is_the_reference_loop(cuda):
    """`algo.attach_env(env, ..., drive_loader=True)`, then the reference's loop VERBATIM -- `for batch in env.train_dataloader:
    loss = algo.update(batch, learn=True); algo.step()` -- iterates handles of the engine's own batches: an epoch is as long as
    the sampler's, and a twin DDPG fed the MATERIALISED handles (`batch["state"]` ...: the rows the engine gathers, rebuilt through
    env.collate_slots) step by step ends with the same four networks bit for bit."""
    import recnn_amd
    n = 23
    env, _ = _bench_env(recnn_amd, cuda, n_users=UPB * 6)
    a = _make_algo(recnn_amd, cuda, env, "bf16")
    env.train_dataloader.planner = a
    assert len(list(zip(range(10 ** 6), env.train_dataloader))) == 6          # one epoch of the sampler: 6 batches of UPB users
    a2 = _make_algo(recnn_amd, cuda, env, "bf16")                             # (same seeds: the same permutations)
    env.train_dataloader.planner = a2
    from recnn_amd.nn import fused
    fused.set_defaults(dtype="bf16", mask_mode="hash", seed=SEED)
    torch.manual_seed(12)
    b = recnn_amd.nn.DDPG(recnn_amd.nn.Actor(1290, 128, 256, 6e-1), recnn_amd.nn.Critic(1290, 128, 256, 54e-2)).to(cuda)
    done = 0
    while done < n:                                                           # epochs of 6 batches, the reference's outer loop
        for batch in env.train_dataloader:
            rows = {k: batch[k] for k in ("state", "action", "reward", "next_state", "done")}   # materialised BEFORE its step runs
            assert rows["state"].shape == (ROWS, 1290)
            a2.update(batch, learn=True); a2.step()
            b.update(rows, learn=True); b.step()
            done += 1
            if done == n:
                break
    a2.flush()
    torch.cuda.synchronize()
    want, got = _snapshot(a2), _snapshot(b)
    for net in want:
        for k in want[net]:
            assert torch.equal(want[net][k], got[net][k]), (net, k)
