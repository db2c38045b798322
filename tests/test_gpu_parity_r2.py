"""Round-2 parity additions (VERDICT r1 "what's weak" 2, 5, 12 and ADVICE r1):

  * the B=32 real-reference fixture is checked through stored per-tensor element samples (a transposed / permuted
    update passes an abs().sum() check; it does not pass this one);
  * a frozen-critic run (value lr = 0): nothing is amplified by Adam's eps regime on the value side, the whole
    policy side is compared with the oracle at 1e-5;
  * multi-step `value_update`: the device step / Adam counters advance (fresh dropout masks, right bias correction);
  * a broken in-launch hand-off is REPORTED (RECNN_E_STATE), not silently computed through;
  * two ranks on one GPU, real HIP engines, gloo all-reduce: 2 x 1024 rows == 1 x 2048 rows.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import recnn_oracle as O
from tests.helpers import fro_err, rel_err

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _engine(algo, S, A, H, B, dtype, mask_mode="external", seed=0):
    from recnn_amd.nn.engine import StepEngine
    return StepEngine(algo, S, A, H, B, dtype=dtype, mask_mode=mask_mode, seed=seed)


def _mk_nets(S, A, H):
    def mk(inp, out, init_w):
        l1, l2, l3 = torch.nn.Linear(inp, H), torch.nn.Linear(H, H), torch.nn.Linear(H, out)
        l3.weight.data.uniform_(-init_w, init_w); l3.bias.data.uniform_(-init_w, init_w)
        return {"w1": l1.weight.data.clone(), "b1": l1.bias.data.clone(), "w2": l2.weight.data.clone(),
                "b2": l2.bias.data.clone(), "w3": l3.weight.data.clone(), "b3": l3.bias.data.clone()}
    val = mk(S + A, 1, 54e-2)
    pol = mk(S, A, 6e-1)
    return pol, val


def test_ddpg_full_b32_final_parameters_by_element_samples(cuda, golden_dir):
    """configs[0] (B=32, full-size nets, 12 steps, 2 policy steps): the real reference's final parameters, sampled
    element-wise (`final_sample` of tests/golden/ddpg_full_b32.json = tensor.flatten()[::numel//7][:8]).

    Adam runs with lr = 1e-3, eps = 1e-8 and NO weight decay here, and at B=32 whole gradient rows are exactly zero or
    sit at round-off level (dead relu/dropout units), so a few sampled elements may sit in the eps regime where the
    update is a coin toss of +-lr per step: >= 95 % of the samples must match at rtol 1e-4 and every sample within
    12 steps x lr of the reference (a transposed or permuted update fails both everywhere)."""
    from recnn_amd import _lib as L
    js = json.load(open(os.path.join(golden_dir, "ddpg_full_b32.json")))
    S, A, H, B, steps, seed = js["dims"]
    lr_v, lr_p, wd_v, wd_p = js["hyper"]
    torch.manual_seed(seed)
    pol, val = _mk_nets(S, A, H)
    batches = [{"state": torch.randn(B, S), "action": torch.randn(B, A), "reward": torch.randn(B) * 3.0,
                "next_state": torch.randn(B, S), "done": (torch.rand(B) < 0.1).float()} for _ in range(2)]
    assert abs(float(batches[0]["state"].double().sum()) - js["input_checksum"][0]) < 1e-6
    eng = _engine("ddpg", S, A, H, B, "fp32")
    eng.load_params(L.NET_POLICY, pol); eng.load_params(L.NET_TARGET_POLICY, pol)
    eng.load_params(L.NET_VALUE1, val); eng.load_params(L.NET_TARGET_VALUE1, val)
    eng.set_hyper(policy_opt=dict(lr=lr_p, weight_decay=wd_p), value_opt=dict(lr=lr_v, weight_decay=wd_v))
    eng.set_counters()
    for t in range(steps):
        masks = O.draw_dropout_masks(6, B, H)
        b = batches[t % 2]
        eng.pack_batch(b["state"], b["action"], b["reward"], b["next_state"], b["done"])
        eng.set_external(masks=masks)
        eng.step(B, True, t)
    torch.cuda.synchronize()
    n_all = n_ok = 0
    worst = 0.0
    for tag, ni in (("policy", L.NET_POLICY), ("value", L.NET_VALUE1), ("target_policy", L.NET_TARGET_POLICY),
                    ("target_value", L.NET_TARGET_VALUE1)):
        got = eng.param_views(ni)
        for k in O.PARAM_ORDER:
            v = got[k].detach().cpu().flatten()
            sample = v[:: max(1, v.numel() // 7)][:8].double()
            ref = torch.tensor(js["final_sample"][tag][k], dtype=torch.float64)
            assert sample.shape == ref.shape, (tag, k)
            dev = (sample - ref).abs()
            n_all += ref.numel()
            n_ok += int((dev <= 1e-4 * ref.abs() + 1e-7).sum())
            worst = max(worst, float(dev.max()))
            assert float(dev.max()) <= steps * max(lr_v, lr_p) * 1.01, (tag, k, dev.tolist())
    print(f"b32 element samples: {n_ok}/{n_all} within rtol 1e-4, worst abs deviation {worst:.3e}")
    assert n_ok >= 0.95 * n_all, (n_ok, n_all, worst)


def test_frozen_critic_run_matches_oracle(cuda):
    """B=2048, fp32, 12 steps (two policy steps) with the critic's learning rate at 0: the value side is pure forward
    arithmetic, so every step's losses must sit within 1e-4 of the oracle (measured ~2e-5) and the critic must not move.

    The actor after its FIRST Adam step is checked element-wise: Adam's first step moves every element by lr*sign(g), so
    all elements match except those whose gradient sign is decided by round-off -- one relu gate whose pre-activation sits
    within fp32 round-off of zero (a few per step among 2048 x 256 x 6 units; a thread-count change flips them in the
    reference too) changes ONE row's contribution to a hidden unit's weight gradients and flips the sign of the few
    percent of them that are smaller than that contribution; those elements then differ by exactly 2 lr.  Allowed:
    <= 1.5 % of the elements (measured 0.7 %), none further than 2.02 lr.  After the SECOND Adam step no element-wise bound exists at this
    learning rate (lr = 1e-3 = 3 % of a typical weight: the 2 lr offsets change the actor's outputs by ~1 %, hence every
    later gradient): only the losses and a loose Frobenius bound are asserted there."""
    from recnn_amd import _lib as L
    S, A, H, B = 1290, 128, 256, 2048
    lr = 1e-3
    torch.manual_seed(0)
    actor, critic = _mk_nets(S, A, H)
    gen = torch.Generator().manual_seed(1)
    batches = [{"state": torch.randn(B, S, generator=gen), "action": torch.randn(B, A, generator=gen),
                "reward": torch.randn(B, generator=gen) * 3.0, "next_state": torch.randn(B, S, generator=gen),
                "done": (torch.rand(B, generator=gen) < 0.1).float()} for _ in range(2)]
    ost = O.DDPGState.create(O.clone_params(actor), O.clone_params(critic), O.AdamState(lr=lr), O.AdamState(lr=0.0))
    eng = _engine("ddpg", S, A, H, B, "fp32")
    eng.load_params(L.NET_POLICY, actor); eng.load_params(L.NET_TARGET_POLICY, actor)
    eng.load_params(L.NET_VALUE1, critic); eng.load_params(L.NET_TARGET_VALUE1, critic)
    eng.set_hyper(policy_opt=dict(lr=lr), value_opt=dict(lr=0.0))
    eng.set_counters()
    worst = 0.0
    for t in range(12):
        masks = [(torch.rand(B, H, generator=gen) < 0.5).to(torch.uint8) for _ in range(6)]
        b = batches[t % 2]
        ref = O.ddpg_step(ost, b, masks, step=t, learn=True)
        eng.pack_batch(b["state"], b["action"], b["reward"], b["next_state"], b["done"])
        eng.set_external(masks=masks)
        eng.step(B, True, t)
        lo = eng.losses()
        for k in ("value", "policy"):
            worst = max(worst, abs(lo[k] - ref[k]) / (abs(ref[k]) + 1e-6))
        if t == 0:
            n_bad = n_all = 0
            far = 0.0
            got = eng.param_views(L.NET_POLICY)
            for k in O.PARAM_ORDER:
                dev = (got[k].detach().cpu() - ost.policy[k]).abs()
                bad = dev > 1e-4 * ost.policy[k].abs() + 1e-4 * ost.policy[k].pow(2).mean().sqrt()
                n_bad += int(bad.sum())
                n_all += dev.numel()
                far = max(far, float(dev.max()))
            print(f"frozen critic, after Adam step 1: {n_bad}/{n_all} actor elements outside rtol 1e-4, farthest {far / lr:.3f} lr")
            assert n_bad <= 1.5e-2 * n_all, (n_bad, n_all)
            assert far <= 2.02 * lr, far
    fro = max(fro_err(eng.param_views(L.NET_POLICY)[k], ost.policy[k]) for k in O.PARAM_ORDER)
    print(f"frozen critic: worst loss deviation over 12 steps {worst:.2e}; actor Frobenius deviation after 2 Adam steps {fro:.2e}")
    assert worst <= 1e-4, worst
    assert fro <= 5e-2, fro
    for k in O.PARAM_ORDER:                                           # lr = 0: the critic did not move
        assert torch.equal(eng.param_views(L.NET_VALUE1)[k].cpu(), critic[k]), k
        assert rel_err(eng.param_views(L.NET_TARGET_VALUE1)[k], ost.target_value[k]) < 1e-6, k   # t*(1-tau) + p*tau, same operand order


def test_value_update_multi_step_advances_device_counters(cuda):
    """`value_update` never closed the step, so repeated calls reused the dropout masks of step 0 and the Adam
    bias correction of t = 1.  Four calls against the oracle: per-call loss at 1e-4 with the hash masks of steps 0..3,
    and the accumulated parameter DELTA (which a frozen t = 1 correction inflates by up to 1.9x) within 1 %."""
    import recnn_amd
    from recnn_amd import _lib as L
    from recnn_amd.nn import fused
    fused.set_defaults(dtype="fp32", mask_mode="hash", seed=77)
    torch.manual_seed(5)
    ddpg = recnn_amd.nn.DDPG(recnn_amd.nn.Actor(1290, 128, 256, 6e-1), recnn_amd.nn.Critic(1290, 128, 256, 54e-2)).to(cuda)
    B = 96
    gen = torch.Generator().manual_seed(2)
    batch = {"state": torch.randn(B, 1290, generator=gen).to(cuda), "action": torch.randn(B, 128, generator=gen).to(cuda),
             "reward": (torch.randn(B, generator=gen) * 3).to(cuda), "next_state": torch.randn(B, 1290, generator=gen).to(cuda),
             "done": (torch.rand(B, generator=gen) < 0.1).float().to(cuda)}
    ost = O.DDPGState.create(O.params_from_module(ddpg.nets["policy_net"]), O.params_from_module(ddpg.nets["value_net"]),
                             O.AdamState(lr=1e-5, weight_decay=1e-2), O.AdamState(lr=1e-5, weight_decay=1e-2))
    w_init = {k: v.clone() for k, v in ost.value.items()}
    cpu_batch = {k: v.cpu() for k, v in batch.items()}
    for t in range(4):
        loss = recnn_amd.nn.update.value_update(batch, ddpg.params, ddpg.nets, ddpg.optimizers, device=cuda, learn=True, step=t)
        masks = []
        for stream in range(2):
            m = torch.zeros(B, 256, dtype=torch.uint8, device=cuda)
            L.call("recnn_hash_mask_dump", 77, t, stream, B, 256, L.ptr(m), L.current_stream())
            torch.cuda.synchronize()
            masks.append(m.cpu())
        ref = O.ddpg_step(ost, cpu_batch, masks, step=1, learn=True)          # step=1: no policy update
        assert abs(float(loss) - ref["value"]) <= 1e-4 * abs(ref["value"]) + 1e-6, (t, float(loss), ref)
    eng = fused.context_for("ddpg", ddpg.nets).engine
    assert eng.counters() == (4, 0, 4, 0)
    got = O.params_from_module(ddpg.nets["value_net"])
    for k in ("w1", "w2"):
        d_got, d_ref = got[k] - w_init[k], ost.value[k] - w_init[k]
        assert fro_err(d_got, d_ref) < 1e-2, (k, fro_err(d_got, d_ref))
    st = ddpg.optimizers["value_optimizer"].state[ddpg.nets["value_net"].linear1.weight]
    assert int(st["step"]) == 4


@pytest.mark.parametrize("fault", [1, 2])
def test_broken_handoff_is_reported(cuda, fault):
    """The cross-workgroup waits of the fused forward are bounded; a wait that runs out must surface as RECNN_E_STATE at
    the next loss / counter read instead of a silently wrong TD target.  recnn_debug_mlp_fault (csrc/recnn_hip_debug.h) breaks one hand-off on
    purpose: 1 = the layer-1 part flags of the chained target critic are never raised, 2 = Q(s, a) never reaches the
    head.  Afterwards the engine is usable again (flags / slots are back at rest, the error word is cleared)."""
    from recnn_amd import _lib as L
    S, A, H, B = 1290, 128, 256, 256
    torch.manual_seed(3)
    actor, critic = _mk_nets(S, A, H)
    gen = torch.Generator().manual_seed(4)
    b = {"state": torch.randn(B, S, generator=gen), "action": torch.randn(B, A, generator=gen),
         "reward": torch.randn(B, generator=gen), "next_state": torch.randn(B, S, generator=gen),
         "done": (torch.rand(B, generator=gen) < 0.1).float()}
    eng = _engine("ddpg", S, A, H, B, "bf16", mask_mode="none", seed=9)     # no dropout: repeated evaluations are identical
    eng.load_params(L.NET_POLICY, actor); eng.load_params(L.NET_TARGET_POLICY, actor)
    eng.load_params(L.NET_VALUE1, critic); eng.load_params(L.NET_TARGET_VALUE1, critic)
    eng.set_hyper(policy_opt=dict(lr=1e-3), value_opt=dict(lr=1e-3))
    eng.set_counters()
    eng.pack_batch(b["state"], b["action"], b["reward"], b["next_state"], b["done"])
    eng.step(B, False, 1)
    good = eng.losses()
    assert np.isfinite(good["value"])
    try:
        # (the hand-offs belong to the fused row-panel forward: a suite run with RECNN_SPLIT_FWD=2 evaluates eager steps with
        # the split forward, which has none -- this test is about csrc/mlps.hip)
        eng.set_tuning(split_fwd=1)
        L.load().recnn_debug_mlp_fault(fault)
        eng.step(B, False, 1)
        with pytest.raises(L.RecnnHipError, match="hand-off timed out"):
            eng.losses()
    finally:
        L.load().recnn_debug_mlp_fault(0)
        eng.set_tuning()
    eng.step(B, False, 1)
    again = eng.losses()                                   # the error word was cleared, flags and slots are at rest
    assert again["value"] == good["value"] and again["policy"] == good["policy"]
    assert eng.counters()[0] == 3


@pytest.mark.parametrize("dtype,mode", [("fp32", "graphs"), ("bf16", "graphs"), ("bf16", "eager")])
def test_two_ranks_on_one_gpu_equal_one_rank(cuda, dtype, mode, tmp_path):
    """World size 2 with the REAL HIP engine (both ranks on this GPU, gloo all-reduce of the flat gradient arenas where
    RCCL would run): 2 x 1024 rows with the matching slices of the global dropout masks == 1 x 2048 rows.  Per-row
    arithmetic is identical; only the fp32 summation order of the gradient reductions differs."""
    port = 29600 + (os.getpid() % 200)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "dp2_worker.py"), str(tmp_path), dtype, mode]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + "\n" + r.stderr[-3000:]
    js = json.load(open(os.path.join(tmp_path, f"dp2_{dtype}_{mode}.json")))
    assert js["world"] == 2 and js["rows_per_rank"] == 1024
    assert js["replica_gap"] == 0.0                                    # replicas bit-identical, no broadcast
    # Adam's sign-like first steps amplify the fp32 summation-order difference of the two reductions (see
    # test_frozen_critic_run_matches_oracle): losses to north_star's 1e-4 in fp32, parameters in Frobenius norm
    ltol = 1e-4 if dtype == "fp32" else 2e-3
    for t, (a, b) in enumerate(zip(js["dp_losses"], js["ref_losses"])):
        for x, y in zip(a, b):
            assert abs(x - y) <= ltol * max(abs(y), 1.0), (t, a, b)
    ptol = 1e-3 if dtype == "fp32" else 1e-2
    for name, e in js["param_err"].items():
        assert e["fro"] <= ptol, (name, e)
    print("dp2", dtype, mode, json.dumps(js["param_err"]))
