"""csrc/dwadam.hip (round 6): the critics' weight-gradient GEMMs with the optimizer in their epilogue -- ONE launch -- against the two launches
it replaces (gemm_dw_dma_kernel's split-batch slabs + apply_kernel's slab sum and optimizer pass).

recnn/nn/update/misc.py:42-44 (value_loss.backward(); optimizer.step()), ddpg.py:95-97 / td3.py:136-141 (soft_update of the target critics
on policy steps).  Same MFMA products in the same order per accumulator, the slabs' batch partition kept as the consumer waves' ranges,
apply_kernel's summation order, the same opt_elem: the parameters, both moments, the gradient arenas, the target networks and (through the
next step's forward) the bf16 compute shadows must agree BIT FOR BIT, for Adam and for Ranger (Lookahead syncs included)."""
import pytest
import torch

from tests.test_gpu_engine import _engine, _init_nets, _rand_batch

pytestmark = pytest.mark.gpu
S, A, H = 1290, 128, 256


def _run(algo, B, fuse, steps, L, opt="adam", policy_every=2, dtype="bf16"):
    td3 = algo == "td3"
    actor, critics = _init_nets(8, S, A, H, 2 if td3 else 1)
    eng = _engine(algo, S, A, H, B, dtype, mask_mode="hash", seed=17)
    eng.set_tuning(split_fwd=2, dw_fuse=fuse)      # split forward for eager steps too: the tail's backward tensors feed either path
    nets = [(L.NET_POLICY, actor), (L.NET_TARGET_POLICY, actor), (L.NET_VALUE1, critics[0]), (L.NET_TARGET_VALUE1, critics[0])]
    if td3:
        nets += [(L.NET_VALUE2, critics[1]), (L.NET_TARGET_VALUE2, critics[1])]
    for ni, p in nets:
        eng.load_params(ni, p)
    o = dict(kind=opt, lr=1e-3, weight_decay=1e-2)
    eng.set_hyper(policy_opt=dict(o), value_opt=dict(o), policy_every=policy_every, soft_tau=0.05)
    eng.set_counters()
    out = []
    gen = torch.Generator().manual_seed(31)
    for t in range(steps):
        batch = _rand_batch(B, S, A, gen)
        eng.pack_batch(batch["state"], batch["action"], batch["reward"], batch["next_state"], batch["done"])
        eng.step(B, True, t)
        torch.cuda.synchronize()
        rec = dict(loss=eng.losses(), g={ni: eng.grads[ni].clone() for ni in eng.value_nets()}, p={ni: eng.params[ni].clone() for ni, _ in nets},
                   m={ni: eng.adam_m[ni].clone() for ni in eng.value_nets()}, v={ni: eng.adam_v[ni].clone() for ni in eng.value_nets()},
                   slow={ni: eng.slow[ni].clone() for ni in eng.value_nets() if ni in eng.slow})
        out.append(rec)
    return out, eng


@pytest.mark.parametrize("algo,B,opt,steps", [("ddpg", 2048, "adam", 5), ("ddpg", 1024, "adam", 3), ("td3", 4096, "adam", 3),
                                              ("ddpg", 2048, "ranger", 13), ("ddpg", 256, "adam", 3)])
def test_fused_dw_adam_equals_dw_launch_plus_optimizer_launch(cuda, algo, B, opt, steps):
    from recnn_amd import _lib as L
    ref, _ = _run(algo, B, 0, steps, L, opt)
    new, eng = _run(algo, B, 1, steps, L, opt)
    prof = [n for n, _, _ in eng.profile(B, policy=False, n_steps=1)]
    if B >= 1024:
        assert "dwadam_critic" in prof and "dw_critic" not in prof and "adam_critic" not in prof, prof
    else:     # 256 rows: 2 slabs, below the tile plan -- the engine keeps the two launches on its own
        assert "dwadam_critic" not in prof and "dw_critic" in prof, prof
    for t, (a, b) in enumerate(zip(ref, new)):
        assert a["loss"] == b["loss"], (t, a["loss"], b["loss"])
        for key in ("g", "p", "m", "v", "slow"):
            for ni in a[key]:
                d = (a[key][ni] - b[key][ni]).abs().max().item()
                assert torch.equal(a[key][ni], b[key][ni]), (t, key, ni, d, int((a[key][ni] != b[key][ni]).sum()))


def test_two_launch_path_is_kept_where_the_tile_plan_does_not_fit(cuda):
    """333 rows (not a multiple of 256): the engine falls back to dW + optimizer launches without being asked."""
    from recnn_amd import _lib as L
    _, eng = _run("ddpg", 333, 1, 2, L)
    prof = [n for n, _, _ in eng.profile(333, policy=False, n_steps=1)]
    assert "dwadam_critic" not in prof and "dw_critic" in prof, prof


@pytest.mark.parametrize("algo,B,opt,steps", [("ddpg", 2048, "adam", 5), ("td3", 4096, "adam", 3), ("ddpg", 1024, "ranger", 7)])
def test_fused_dw_adam_in_split_bf16_equals_the_two_launches(cuda, algo, B, opt, steps):
    """The split-bf16 (bf16x3) form of dw_adam_kernel: one 16 x 16 block per consumer wave over all slabs, three MFMAs per product in
    x3_dw_kernel's order, the slabs summed per lane in apply_kernel's order, split shadows written from the epilogue -- against
    x3_dw_kernel + apply_kernel, bit for bit (parameters, moments, gradient arenas, targets; the shadows through the next step's forward)."""
    from recnn_amd import _lib as L
    ref, _ = _run(algo, B, 0, steps, L, opt, dtype="bf16x3")
    new, eng = _run(algo, B, 1, steps, L, opt, dtype="bf16x3")
    prof = [n for n, _, _ in eng.profile(B, policy=False, n_steps=1)]
    assert "dwadam_critic" in prof and "dw_critic" not in prof and "adam_critic" not in prof, prof
    for t, (a, b) in enumerate(zip(ref, new)):
        assert a["loss"] == b["loss"], (t, a["loss"], b["loss"])
        for key in ("g", "p", "m", "v", "slow"):
            for ni in a[key]:
                d = (a[key][ni] - b[key][ni]).abs().max().item()
                assert torch.equal(a[key][ni], b[key][ni]), (t, key, ni, d, int((a[key][ni] != b[key][ni]).sum()))
