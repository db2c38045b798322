"""Ranger (RAdam + Lookahead), the reference's DEFAULT optimizer shape (torch_optimizer.Ranger, recnn/nn/algo.py:84-89):
fused HIP arithmetic (`recnn_ranger_flat`, the engine's optimizer pass) against

  * the same algorithm in plain torch ops (`Ranger.reference_step`),
  * torch.optim.RAdam + a three-line Lookahead where the two coincide (weight_decay = 0): same rectification term, same
    N_sma > 5 switch, eps outside the square root -- this is what pins the restatement,
  * the generic (non-fused) optimizer path of the update functions.

`torch_optimizer` itself is absent and un-pinned, so parity with IT stays unpinned."""
import numpy as np
import pytest
import torch

from tests.helpers import rel_err

pytestmark = pytest.mark.gpu


def test_ranger_kernel_matches_torch_ops_restatement(cuda):
    import recnn_amd
    torch.manual_seed(0)
    w0 = torch.randn(300, 70, device=cuda)
    grads = [torch.randn(300, 70, device=cuda) * 1e-2 for _ in range(14)]       # crosses N_sma = 5 and two Lookahead syncs
    pa, pb = torch.nn.Parameter(w0.clone()), torch.nn.Parameter(w0.clone())
    oa = recnn_amd.optim.Ranger([pa], lr=3e-3, weight_decay=1e-2)
    ob = recnn_amd.optim.Ranger([pb], lr=3e-3, weight_decay=1e-2)
    for g in grads:
        pa.grad, pb.grad = g.clone(), g.clone()
        oa.step()
        ob.reference_step()
        assert rel_err(pa, pb) < 1e-6
    for k in ("exp_avg", "exp_avg_sq", "slow_buffer"):
        assert rel_err(oa.state[pa][k], ob.state[pb][k]) < 1e-6, k
    assert oa.state[pa]["step"] == 14
    cfg = recnn_amd.optim.fused_config(oa)
    assert cfg["kind"] == "ranger" and cfg["k"] == 6 and cfg["alpha"] == 0.5 and cfg["beta1"] == 0.95 and cfg["eps"] == 1e-5


def test_ranger_pinned_to_torch_radam_plus_lookahead(cuda):
    """weight_decay = 0: RAdam's update is identical in torch.optim.RAdam and in the published Ranger
    (rectified: lr * m_hat * r_t * sqrt(1 - b2^t) / (sqrt(v) + eps); else lr * m_hat), Lookahead(k, alpha) on top."""
    import recnn_amd
    torch.manual_seed(1)
    w0 = torch.randn(257, 33, device=cuda)
    grads = [torch.randn(257, 33, device=cuda) * 3e-2 for _ in range(20)]
    pa, pb = torch.nn.Parameter(w0.clone()), torch.nn.Parameter(w0.clone())
    oa = recnn_amd.optim.Ranger([pa], lr=1e-3, alpha=0.5, k=6, betas=(0.95, 0.999), eps=1e-5, weight_decay=0)
    ob = torch.optim.RAdam([pb], lr=1e-3, betas=(0.95, 0.999), eps=1e-5, weight_decay=0)
    slow = w0.clone()
    for t, g in enumerate(grads, 1):
        pa.grad, pb.grad = g.clone(), g.clone()
        oa.step()
        ob.step()
        if t % 6 == 0:
            with torch.no_grad():
                slow += 0.5 * (pb.data - slow)
                pb.data.copy_(slow)
        assert rel_err(pa, pb) < 2e-6, t


def _ci_batch(dev, B=96):
    g = torch.Generator().manual_seed(0)
    return {"state": torch.randn(B, 1290, generator=g).to(dev), "action": torch.randn(B, 128, generator=g).to(dev),
            "reward": (torch.randn(B, generator=g) * 3).to(dev), "next_state": torch.randn(B, 1290, generator=g).to(dev),
            "done": (torch.rand(B, generator=g) < 0.1).float().to(dev)}


@pytest.mark.parametrize("algo_name", ["ddpg", "td3"])
def test_default_optimizer_is_fused_ranger_and_matches_generic_path(cuda, algo_name):
    """The facades build Ranger(lr=1e-5, weight_decay=1e-2) like the reference; inside the update functions it runs in the
    engine's optimizer pass.  Same run with a Ranger SUBCLASS (not recognised -> gradients handed to opt.step(), the
    generic path): parameters, moments and slow weights agree."""
    import recnn_amd
    from recnn_amd.nn import algo as algo_mod, fused
    algo_mod.set_default_optimizer("ranger")
    fused.set_defaults(dtype="fp32", mask_mode="hash", seed=5)

    class MyRanger(recnn_amd.optim.Ranger):
        pass

    batch = _ci_batch(cuda)
    outs = []
    for mode in ("fused", "generic"):
        torch.manual_seed(3)
        if algo_name == "ddpg":
            a = recnn_amd.nn.DDPG(recnn_amd.nn.Actor(1290, 128, 256, 6e-1), recnn_amd.nn.Critic(1290, 128, 256, 54e-2)).to(cuda)
            a.params["policy_step"] = 3
            vkeys = ("value_optimizer",)
        else:
            a = recnn_amd.nn.TD3(recnn_amd.nn.Actor(1290, 128, 256, 6e-1), recnn_amd.nn.Critic(1290, 128, 256, 54e-2),
                                 recnn_amd.nn.Critic(1290, 128, 256, 54e-2)).to(cuda)
            a.params["policy_update"] = 3
            vkeys = ("value_optimizer1", "value_optimizer2")
        for k, o in a.optimizers.items():
            assert type(o) is recnn_amd.optim.Ranger and o.defaults["lr"] == 1e-5 and o.defaults["weight_decay"] == 1e-2, k
        for k in a.optimizers:        # a learning rate at which 14 steps move the weights visibly
            net = {"policy_optimizer": "policy_net", "value_optimizer": "value_net", "value_optimizer1": "value_net1",
                   "value_optimizer2": "value_net2"}[k]
            cls = recnn_amd.optim.Ranger if mode == "fused" else MyRanger
            a.optimizers[k] = cls(a.nets[net].parameters(), lr=1e-3, weight_decay=1e-2)
        losses = []
        for t in range(14):
            losses.append(a.update(batch, learn=True))
            a.step()
        torch.cuda.synchronize()
        vnet = "value_net" if algo_name == "ddpg" else "value_net1"
        st_v = a.optimizers[vkeys[0]].state[a.nets[vnet].linear1.weight]
        st_p = a.optimizers["policy_optimizer"].state[a.nets["policy_net"].linear2.weight]
        outs.append((losses, {n: {k: v.detach().clone() for k, v in m.state_dict().items()} for n, m in a.nets.items()},
                     {k: st_v[k].detach().clone() for k in ("exp_avg", "exp_avg_sq", "slow_buffer")}, int(st_v["step"]),
                     {k: st_p[k].detach().clone() for k in ("exp_avg", "exp_avg_sq", "slow_buffer")}, int(st_p["step"])))
    assert outs[0][3] == outs[1][3] == 14 and outs[0][5] == outs[1][5] == 5          # policy steps 0, 3, 6, 9, 12
    for la, lb in zip(outs[0][0], outs[1][0]):
        for k in la:
            assert abs(la[k] - lb[k]) <= 1e-5 * max(abs(lb[k]), 1.0), (la, lb)
    for n in outs[0][1]:
        for k in outs[0][1][n]:
            assert rel_err(outs[0][1][n][k], outs[1][1][n][k]) < 1e-5, (n, k)
    for j in (2, 4):
        for k in outs[0][j]:
            assert rel_err(outs[0][j][k], outs[1][j][k]) < 1e-5, (j, k)


def test_run_graphs_with_default_ranger(cuda):
    """Algo.run (hipGraph replay, device-side step counters deciding the Lookahead syncs) == the update loop, bit for bit."""
    import recnn_amd
    from recnn_amd.nn import algo as algo_mod, fused
    from tests.test_gpu_bench_shape import _bench_env, ROWS, UPB
    algo_mod.set_default_optimizer("ranger")
    env, table = _bench_env(recnn_amd, cuda, n_users=30 * UPB)
    outs = []
    for mode in ("run", "loop"):
        fused.set_defaults(dtype="bf16", mask_mode="hash", seed=17)
        torch.manual_seed(12)
        ddpg = recnn_amd.nn.DDPG(recnn_amd.nn.Actor(1290, 128, 256, 6e-1), recnn_amd.nn.Critic(1290, 128, 256, 54e-2)).to(cuda)
        assert type(ddpg.optimizers["policy_optimizer"]) is recnn_amd.optim.Ranger
        torch.manual_seed(99)
        ddpg.attach_env(env, rows_per_batch=ROWS, users_per_batch=UPB)
        ctx = ddpg._fused_ctx
        n = 27
        if mode == "run":
            ddpg.run(5)
            ddpg.run(n - 5)
        else:
            perm = ctx.perm.cpu().numpy()
            for i in range(n):
                ddpg.update(env.collate_users([int(u) for u in perm[i * UPB:(i + 1) * UPB]]), learn=True)
                ddpg.step()
        torch.cuda.synchronize()
        st = ddpg.optimizers["value_optimizer"].state[ddpg.nets["value_net"].linear1.weight]
        outs.append(({nn: {k: v.detach().clone() for k, v in m.state_dict().items()} for nn, m in ddpg.nets.items()},
                     st["slow_buffer"].detach().clone(), int(st["step"])))
    assert outs[0][2] == outs[1][2] == 27
    assert torch.equal(outs[0][1], outs[1][1])
    for nn in outs[0][0]:
        for k in outs[0][0][nn]:
            assert torch.equal(outs[0][0][nn][k], outs[1][0][nn][k]), (nn, k)
    assert not torch.equal(outs[0][1], outs[0][0]["value_net"]["linear1.weight"])     # fast and slow weights differ between syncs


@pytest.mark.parametrize("opt_name", ["ranger", "adam"])
@pytest.mark.parametrize("bf16", [False, True])
@pytest.mark.parametrize("cols", [2049, 2050, 2052])
def test_optimizer_rewrites_the_cached_compute_layout_in_the_same_pass(cuda, opt_name, bf16, cols):
    """recnn_*_flat_shadow: the optimizer kernel also writes the row-padded (optionally bf16) copy of a catalogue-sized weight that
    the GEMM kernels read (recnn_amd.nn.functional `_derived_of` layouts), so the next forward does not spend a conversion pass over
    it.  The parameter and optimizer state equal the plain step bit for bit; the copy equals a rebuild from the new parameter, bit
    for bit (fp32: the value; bf16: round to nearest even), padding columns untouched; the cache reports it current."""
    import recnn_amd
    from recnn_amd.nn import functional as Fh
    torch.manual_seed(1)
    # > 4M elements; the flat pass takes 16 bytes per lane and writes the copy per quad: element by element (odd column count: quads
    # straddle rows and a two-element tail is left), as two pairs (2050: the catalogue-sized weights of REINFORCE are of this kind) or
    # as one store (2052).
    rows, ld = 2050, 2112
    w0 = torch.randn(rows, cols, device=cuda)
    make = (lambda ps: recnn_amd.optim.Ranger(ps, lr=1e-3, weight_decay=1e-2)) if opt_name == "ranger" else \
        (lambda ps: recnn_amd.optim.Adam(ps, lr=1e-3, weight_decay=1e-2))
    pa, pb = torch.nn.Parameter(w0.clone()), torch.nn.Parameter(w0.clone())
    oa, ob = make([pa]), make([pb])
    dt = torch.bfloat16 if bf16 else torch.float32
    kind = "bf16_padded" if bf16 else "padded"

    def build(w):
        t = torch.full((rows + 2, ld), 7.0, dtype=dt, device=cuda)      # (7 in the padding: the kernel must not touch it)
        t[:rows, :cols] = w
        return t
    for it in range(7):                                      # crosses a Lookahead sync for Ranger
        shadow = Fh._derived_of(pa, kind, build)             # current before the step (a forward would have made it)
        assert Fh.shadow_target(pa) is not None and Fh.shadow_target(pa)[1] is shadow
        assert Fh.shadow_target(pb) is None                  # nothing cached for the twin: plain step
        g = torch.randn(rows, cols, device=cuda) * 1e-2
        pa.grad, pb.grad = g.clone(), g.clone()
        oa.step()
        ob.step()
        torch.cuda.synchronize()
        assert torch.equal(pa, pb), it
        hit = Fh._derived_of(pa, kind, lambda w: (_ for _ in ()).throw(AssertionError("the cached copy was not reported current")))
        assert hit is shadow
        assert torch.equal(hit[:rows, :cols], pa.detach().to(dt)), it
        assert bool((hit[:rows, cols:] == 7).all()) and bool((hit[rows:] == 7).all())
    for k in ("exp_avg", "exp_avg_sq"):
        assert torch.equal(oa.state[pa][k], ob.state[pb][k]), k


@pytest.mark.parametrize("opt_name", ["ranger", "adam"])
def test_flat_pass_with_16_byte_lanes_equals_the_4_byte_form(cuda, opt_name):
    """The flat optimizer kernels take 16 bytes per lane when every array is 16-byte aligned and 4 bytes otherwise (a parameter that is a
    view at an odd offset): element by element the same arithmetic, so the same bits -- parameter and state, through a Lookahead sync."""
    import recnn_amd
    torch.manual_seed(2)
    n = 1_000_003                                            # (n % 4 = 3: the 16-byte form leaves a tail)
    w0 = torch.randn(n, device=cuda)
    base = torch.zeros(n + 1, device=cuda)
    base[1:] = w0
    pa, pb = torch.nn.Parameter(w0.clone()), torch.nn.Parameter(base[1:])
    assert pa.data_ptr() % 16 == 0 and pb.data_ptr() % 16 == 4
    make = (lambda ps: recnn_amd.optim.Ranger(ps, lr=1e-3, weight_decay=1e-2)) if opt_name == "ranger" else \
        (lambda ps: recnn_amd.optim.Adam(ps, lr=1e-3, weight_decay=1e-2))
    oa, ob = make([pa]), make([pb])
    for it in range(7):
        g = torch.randn(n, device=cuda) * 1e-2
        pa.grad, pb.grad = g.clone(), g.clone()
        oa.step()
        ob.step()
        assert torch.equal(pa.detach(), pb.detach()), it
    for k in ("exp_avg", "exp_avg_sq"):
        assert torch.equal(oa.state[pa][k], ob.state[pb][k]), k
