"""The split forward (csrc/l1gemm.hip + csrc/mlpt.hip, round 3) against the fused row-panel kernel (csrc/mlps.hip).

Both compute recnn/nn/models.py:66-73 / :207-213 (the three Linear + relu + dropout layers), recnn/nn/update/misc.py:6-7,33-39
(TD target, MSE) and the backward of the critic's last two layers with the same per-element arithmetic: k ascending in
MFMA steps of 32, [state | action] contracted state-first for the target critics, unit backward tensors rounded to bf16 and
THEN multiplied by the per-row loss seed -- so every buffer of a step, the gradients and the parameters after several
steps (policy steps included) must agree BIT FOR BIT.  What differs is where the work runs (tiled layer-1 GEMMs over the
whole machine, 128-192 KB tails, frozen networks first so that no workgroup waits for another).
"""
import pytest
import torch

from tests.test_gpu_engine import _engine, _init_nets, _rand_batch

pytestmark = pytest.mark.gpu
S, A, H = 1290, 128, 256
BUFS = ("next_action", "gen_action", "expected", "target_q", "q1", "delta1", "critic1_h1", "critic1_h2", "actor_h1", "actor_h2")


def _run(algo, B, split, mask_mode, steps, L, learn_last=True):
    td3 = algo == "td3"
    actor, critics = _init_nets(8, S, A, H, 2 if td3 else 1)
    batch = _rand_batch(B, S, A, torch.Generator().manual_seed(31))
    eng = _engine(algo, S, A, H, B, "bf16", mask_mode=mask_mode, seed=17)
    eng.set_tuning(split_fwd=split)          # per-engine: nothing process-wide is touched
    nets = [(L.NET_POLICY, actor), (L.NET_TARGET_POLICY, actor), (L.NET_VALUE1, critics[0]), (L.NET_TARGET_VALUE1, critics[0])]
    if td3:
        nets += [(L.NET_VALUE2, critics[1]), (L.NET_TARGET_VALUE2, critics[1])]
    for ni, p in nets:
        eng.load_params(ni, p)
    eng.set_hyper(policy_opt=dict(lr=1e-3, weight_decay=1e-2), value_opt=dict(lr=1e-3, weight_decay=1e-2), policy_every=2)
    eng.set_counters()
    eng.pack_batch(batch["state"], batch["action"], batch["reward"], batch["next_state"], batch["done"])
    out = []
    for t in range(steps):
        eng.step(B, True, t)
        torch.cuda.synchronize()
        rec = dict(loss=eng.losses(), bufs={n: eng.buffer(n, B).clone() for n in BUFS},
                   g={ni: eng.grads[ni].clone() for ni in eng.value_nets()},
                   p={ni: eng.params[ni].clone() for ni, _ in nets})
        if td3:
            rec["bufs"]["q2"] = eng.buffer("q2", B).clone()
        out.append(rec)
    eng.step(B, False, steps)                 # a learn=False evaluation as well
    out.append(dict(loss=eng.losses(), bufs={n: eng.buffer(n, B).clone() for n in ("expected", "q1", "gen_action")}, g={}, p={}))
    return out


@pytest.mark.parametrize("algo,B,mask_mode", [("ddpg", 2048, "hash"), ("ddpg", 333, "hash"), ("td3", 1024, "hash"), ("ddpg", 77, "none"),
                                              ("ddpg", 4100, "hash")])
def test_split_forward_equals_fused_row_panel_forward(cuda, algo, B, mask_mode):
    from recnn_amd import _lib as L
    ref = _run(algo, B, 0, mask_mode, 3, L)
    new = _run(algo, B, 2, mask_mode, 3, L)      # 2 = split forward for eager steps too
    for t, (a, b) in enumerate(zip(ref, new)):
        for n in a["bufs"]:
            assert torch.equal(a["bufs"][n], b["bufs"][n]), (t, n, (a["bufs"][n] - b["bufs"][n]).abs().max().item())
        assert a["loss"] == b["loss"], (t, a["loss"], b["loss"])
        for ni in a["g"]:
            assert torch.equal(a["g"][ni], b["g"][ni]), (t, "grad", ni, (a["g"][ni] - b["g"][ni]).abs().max().item())
        for ni in a["p"]:
            assert torch.equal(a["p"][ni], b["p"][ni]), (t, "param", ni, (a["p"][ni] - b["p"][ni]).abs().max().item())


def test_two_engines_with_different_schedules_coexist(cuda):
    """The tuning is per engine (recnn_engine_set_tuning): an engine on the fused row-panel forward and one on the split forward,
    created side by side and stepped ALTERNATELY in one process, stay bit-identical -- no process-wide schedule state."""
    from recnn_amd import _lib as L
    B = 512
    actor, critics = _init_nets(8, S, A, H, 1)
    batch = _rand_batch(B, S, A, torch.Generator().manual_seed(31))
    engs = []
    for split in (0, 2):
        eng = _engine("ddpg", S, A, H, B, "bf16", mask_mode="hash", seed=17)
        eng.set_tuning(split_fwd=split, dw_dma=2 if split == 0 else 3)
        for ni, p in ((L.NET_POLICY, actor), (L.NET_TARGET_POLICY, actor), (L.NET_VALUE1, critics[0]), (L.NET_TARGET_VALUE1, critics[0])):
            eng.load_params(ni, p)
        eng.set_hyper(policy_opt=dict(lr=1e-3), value_opt=dict(lr=1e-3), policy_every=2)
        eng.set_counters()
        eng.pack_batch(batch["state"], batch["action"], batch["reward"], batch["next_state"], batch["done"])
        engs.append(eng)
    assert engs[0].tuning.split_fwd == 0 and engs[1].tuning.split_fwd == 2
    for t in range(4):
        for eng in engs:                          # interleaved: each launch sequence is planned from its own engine's tuning
            eng.step(B, True, t)
        torch.cuda.synchronize()
        assert engs[0].losses() == engs[1].losses(), t
    for ni in (L.NET_POLICY, L.NET_VALUE1, L.NET_TARGET_POLICY, L.NET_TARGET_VALUE1):
        assert torch.equal(engs[0].params[ni], engs[1].params[ni]), ni
    got = L.EngineTuning()
    L.call("recnn_engine_get_tuning", engs[1].handle, got)
    assert got.split_fwd == 2 and got.dw_dma == 3


@pytest.mark.parametrize("algo,B", [("ddpg", 2048), ("td3", 1024), ("ddpg", 96), ("ddpg", 333)])
def test_half_panel_tail_equals_whole_panel_tail(cuda, algo, B):
    """tuning.tail_half (round 6): the learning critic's tail launch on 16-row panels -- the small tensors' panel sums and the value-loss
    partials then arrive as HALF-panel sums and are consumed in pairs ((c0 + c1) + (c2 + c3), optim_dev.h / head.hip) -- against
    32-row panels: every buffer, gradient and parameter bit for bit (333 rows: not whole pairs, the engine keeps 32-row panels)."""
    from recnn_amd import _lib as L

    def run(half):
        from recnn_amd._tune import set_default_tuning
        set_default_tuning(tail_half=half)
        try:
            return _run(algo, B, 2, "hash", 4, L)
        finally:
            set_default_tuning(tail_half=None)
    ref, new = run(0), run(1)
    for t, (a, b) in enumerate(zip(ref, new)):
        for n in a["bufs"]:
            assert torch.equal(a["bufs"][n], b["bufs"][n]), (t, n, (a["bufs"][n] - b["bufs"][n]).abs().max().item())
        assert a["loss"] == b["loss"], (t, a["loss"], b["loss"])
        for ni in a["g"]:
            assert torch.equal(a["g"][ni], b["g"][ni]), (t, "grad", ni, (a["g"][ni] - b["g"][ni]).abs().max().item())
        for ni in a["p"]:
            assert torch.equal(a["p"][ni], b["p"][ni]), (t, "param", ni, (a["p"][ni] - b["p"][ni]).abs().max().item())
