"""GPU tests of the drop-in Python API (recnn_amd.nn / recnn_amd.data / recnn_amd.utils).

The first three tests are the reference's own CI tests (.circleci/tests/learning.py:24-92) re-expressed
against this package on the GPU: same fixtures, same assertions.
"""
import numpy as np
import pytest
import torch

from oracle import recnn_oracle as O
from tests.helpers import fro_err, make_store, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture()
def recnn(cuda):
    import recnn_amd
    from recnn_amd.nn import fused
    fused.set_defaults(dtype="fp32", mask_mode="hash", seed=11)
    return recnn_amd


def _ci_batch(dev):
    g = torch.Generator().manual_seed(0)
    return {"state": torch.randn(10, 1290, generator=g).to(dev), "action": torch.randn(10, 128, generator=g).to(dev),
            "reward": torch.randn(10, 1, generator=g).to(dev), "next_state": torch.randn(10, 1290, generator=g).to(dev),
            "done": torch.randn(10, 1, generator=g).to(dev)}           # Gaussian "done", as in learning.py:11


def _check_loss_and_networks(loss, nets):
    assert loss["value"] > 0 and loss["policy"] != 0 and loss["step"] == 0
    for name, netw in nets.items():
        assert netw.training == ("target" not in name)


def test_recommendation(recnn, cuda):
    value_net = recnn.nn.Critic(1290, 128, 256, 54e-2).to(cuda)
    policy_net = recnn.nn.Actor(1290, 128, 256, 6e-1).to(cuda)
    state = _ci_batch(cuda)["state"]
    recommendation = policy_net(state)
    value = value_net(state, recommendation)
    assert recommendation.std() > 0 and recommendation.mean() != 0
    assert value.std() > 0


@pytest.mark.parametrize("opt_kind", ["fused_adam", "torch_radam"])
def test_update_function(recnn, cuda, opt_kind):
    value_net = recnn.nn.Critic(1290, 128, 256, 54e-2).to(cuda)
    policy_net = recnn.nn.Actor(1290, 128, 256, 6e-1).to(cuda)
    target_value_net = recnn.nn.Critic(1290, 128, 256).to(cuda)
    target_policy_net = recnn.nn.Actor(1290, 128, 256).to(cuda)
    target_policy_net.eval()
    target_value_net.eval()
    recnn.utils.soft_update(value_net, target_value_net, soft_tau=1.0)
    recnn.utils.soft_update(policy_net, target_policy_net, soft_tau=1.0)
    for a, b in zip(value_net.parameters(), target_value_net.parameters()):
        assert torch.equal(a, b)
    if opt_kind == "fused_adam":
        mk = lambda p: recnn.optim.Adam(p, lr=1e-5, weight_decay=1e-2)
    else:   # learning.py:51-52 uses RAdam(lr=1e-5, weight_decay=1e-2): any torch optimizer must work
        mk = lambda p: torch.optim.RAdam(p, lr=1e-5, weight_decay=1e-2)
    nets = {"value_net": value_net, "target_value_net": target_value_net, "policy_net": policy_net,
            "target_policy_net": target_policy_net}
    optimizer = {"policy_optimizer": mk(policy_net.parameters()), "value_optimizer": mk(value_net.parameters())}
    params = {"gamma": 0.99, "min_value": -10, "max_value": 10, "policy_step": 10, "soft_tau": 0.001}
    debug = {}
    batch = _ci_batch(cuda)
    loss = recnn.nn.update.ddpg_update(batch, params, nets, optimizer, cuda, debug, recnn.utils.misc.DummyWriter(), step=0)
    _check_loss_and_networks(loss, nets)
    assert debug["next_action"].shape == (10, 128) and debug["gen_action"].shape == (10, 128)   # learn=False default
    before = [p.detach().clone() for p in value_net.parameters()]
    loss = recnn.nn.update.ddpg_update(batch, params, nets, optimizer, cuda, debug, recnn.utils.misc.DummyWriter(),
                                       learn=True, step=0)
    _check_loss_and_networks(loss, nets)
    assert any(not torch.equal(a, b) for a, b in zip(before, value_net.parameters()))       # updated in place


def test_algo(recnn, cuda):
    value_net = recnn.nn.Critic(1290, 128, 256, 54e-2)
    policy_net = recnn.nn.Actor(1290, 128, 256, 6e-1)
    ddpg = recnn.nn.DDPG(policy_net, value_net).to(cuda)
    loss = ddpg.update(_ci_batch(cuda), learn=True)
    _check_loss_and_networks(loss, ddpg.nets)
    td3 = recnn.nn.TD3(recnn.nn.Actor(1290, 128, 256, 6e-1), recnn.nn.Critic(1290, 128, 256, 54e-2),
                       recnn.nn.Critic(1290, 128, 256, 54e-2)).to(cuda)
    lo = td3.update(_ci_batch(cuda), learn=True)
    assert lo["value1"] > 0 and lo["value2"] > 0 and lo["policy"] != 0 and lo["step"] == 0
    for name, netw in td3.nets.items():
        assert netw.training == ("target" not in name)


def test_cpu_networks_fail_loudly(recnn):
    from recnn_amd._lib import RecnnHipError
    ddpg = recnn.nn.DDPG(recnn.nn.Actor(1290, 128, 256), recnn.nn.Critic(1290, 128, 256))
    b = {k: v.cpu() for k, v in _ci_batch(torch.device("cpu")).items()}
    with pytest.raises(RecnnHipError):
        ddpg.update(b, learn=True)
    with pytest.raises(RecnnHipError):
        recnn.nn.Actor(1290, 128, 256)(b["state"])


def test_constructor_rng_matches_reference_order(recnn):
    """Equal seeds give the reference's initial weights (models.py:52-57): checked against a replay of that recipe."""
    torch.manual_seed(5)
    a = recnn.nn.Actor(27, 8, 16, 0.3)
    torch.manual_seed(5)
    l1, l2, l3 = torch.nn.Linear(27, 16), torch.nn.Linear(16, 16), torch.nn.Linear(16, 8)
    l3.weight.data.uniform_(-0.3, 0.3)
    l3.bias.data.uniform_(-0.3, 0.3)
    assert torch.equal(a.linear1.weight, l1.weight) and torch.equal(a.linear3.bias, l3.bias)
    assert list(a.state_dict().keys()) == ["linear1.weight", "linear1.bias", "linear2.weight", "linear2.bias",
                                           "linear3.weight", "linear3.bias"]


def test_module_forward_backward_matches_torch(recnn, cuda):
    """Actor/Critic called directly (eval mode: no dropout) against plain torch fp32 autograd."""
    torch.manual_seed(1)
    crit = recnn.nn.Critic(1290, 128, 256, 0.5).to(cuda).eval()
    s = torch.randn(77, 1290, device=cuda)
    a = torch.randn(77, 128, device=cuda, requires_grad=True)
    q = crit(s, a)
    (q * torch.linspace(-1, 1, 77, device=cuda)[:, None]).sum().backward()
    ref_a = a.detach().clone().requires_grad_(True)
    x = torch.cat([s, ref_a], 1)
    w = {k: v.detach().clone().requires_grad_(True) for k, v in crit.named_parameters()}
    h = torch.relu(x @ w["linear1.weight"].t() + w["linear1.bias"])
    h = torch.relu(h @ w["linear2.weight"].t() + w["linear2.bias"])
    qr = h @ w["linear3.weight"].t() + w["linear3.bias"]
    (qr * torch.linspace(-1, 1, 77, device=cuda)[:, None]).sum().backward()
    assert rel_err(q, qr) < 1e-5
    assert rel_err(a.grad, ref_a.grad) < 1e-4
    for k, p in crit.named_parameters():
        assert rel_err(p.grad, w[k].grad) < 1e-4, k
    # train mode: inverted dropout keeps the expectation, drops about half of the hidden units
    act = recnn.nn.Actor(1290, 128, 256, 0.5).to(cuda)
    out_train = torch.stack([act(s) for _ in range(8)])
    assert out_train.std(0).mean() > 0          # masks differ between calls
    act.eval()
    assert torch.equal(act(s), act(s))


def _env(recnn, cuda, rows_per_batch=None, contiguous=False, n_users=40, seed=2, n_train=32):
    items, ratings, table = make_store(n_users=n_users, n_items=500, emb_dim=128, min_len=11, max_len=90, seed=seed)
    user_dict = {100 + 3 * u: {"items": items[u], "ratings": ratings[u]} for u in range(n_users)}
    ids = list(user_dict.keys())
    env = recnn.data.env.FrameEnv.from_user_dict(torch.from_numpy(table), user_dict, ids[:n_train], ids[n_train:], frame_size=10,
                                                 batch_size=5, device=cuda, rows_per_batch=rows_per_batch,
                                                 contiguous=contiguous)
    return env, user_dict, table


@pytest.mark.parametrize("contiguous", [False, True])
def test_frame_env_batches_bit_exact(recnn, cuda, contiguous):
    env, user_dict, table = _env(recnn, cuda, contiguous=contiguous)
    assert len(env.train_dataloader) == 7 and len(env.test_dataloader) == 2     # ceil(32/5), ceil(8/5)
    seen = []
    for epoch in range(2):                                                       # re-iterable, reshuffled
        order = []
        for batch in env.train_dataloader:
            users = batch["meta"]["users"].tolist()
            order += users
            ref = O.frame_batch([user_dict[u]["items"] for u in users], [user_dict[u]["ratings"] for u in users], table, 10)
            for k in ("state", "action", "reward", "next_state", "done"):
                assert np.array_equal(batch[k].cpu().numpy(), ref[k]), k
            assert batch["state"].shape[1] == 1290 and batch["meta"]["sizes"].tolist() == ref["sizes"].tolist()
            assert batch["state"].is_contiguous() == contiguous
        assert sorted(order) == sorted(env.base.train_user_dataset.users)
        seen.append(order)
    assert seen[0] != seen[1]
    b = env.train_batch()
    st, ac, rw, ns, dn = recnn.data.get_base_batch(b, device=cuda)
    assert rw.shape == (st.shape[0], 1) and dn.shape == (st.shape[0], 1) and st.data_ptr() == b["state"].data_ptr()


def test_frame_env_fixed_rows_and_collate_compat(recnn, cuda):
    env, user_dict, table = _env(recnn, cuda, rows_per_batch=64)
    b = env.train_batch()
    assert b["state"].shape == (64, 1290)
    users = b["meta"]["users"].tolist()
    ref = O.frame_batch([user_dict[u]["items"] for u in users], [user_dict[u]["ratings"] for u in users], table, 10, rows=64)
    assert np.array_equal(b["next_state"].cpu().numpy(), ref["next_state"])
    # the reference's collate_fn signature on a list of UserDataset items
    ds = env.base.train_user_dataset
    lst = [ds[i] for i in (3, 1, 4)]
    out = recnn.data.utils.prepare_batch_static_size(lst, env.table, frame_size=10)
    ref = O.frame_batch([x["items"] for x in lst], [x["rates"] for x in lst], table, 10)
    for k in ("state", "action", "reward", "next_state", "done"):
        assert np.array_equal(out[k].cpu().numpy(), ref[k]), k
    # windowed embed function (utils.py:51-81) called directly
    it, rt, sizes = O.frame_windows([x["items"] for x in lst], [x["rates"] for x in lst], 10)
    out2 = recnn.data.batch_tensor_embeddings({"items": torch.from_numpy(it), "ratings": torch.from_numpy(rt),
                                               "sizes": torch.from_numpy(sizes), "users": torch.tensor([0, 1, 2])},
                                              env.table, 10)
    for k in ("state", "action", "reward", "next_state", "done"):
        assert np.array_equal(out2[k].cpu().numpy(), ref[k]), k


def _oracle_state(pol, val, lr, wd_p, wd_v):
    return O.DDPGState.create(O.params_from_module(pol), O.params_from_module(val), O.AdamState(lr=lr, weight_decay=wd_p),
                              O.AdamState(lr=lr, weight_decay=wd_v))


class _PlainAdam(torch.optim.Adam):
    """Not recognised by the fused path: exercises 'any torch optimizer' with Adam arithmetic the oracle knows."""


@pytest.mark.parametrize("mode", ["fused", "generic"])
def test_ddpg_api_training_loop_matches_oracle(recnn, cuda, mode):
    """for batch in env.train_dataloader: algo.update(batch); algo.step()  -- against the oracle, identical masks."""
    from recnn_amd.nn import fused
    env, user_dict, table = _env(recnn, cuda, n_users=60, seed=4)
    torch.manual_seed(3)
    value_net = recnn.nn.Critic(1290, 128, 256, 54e-2)
    policy_net = recnn.nn.Actor(1290, 128, 256, 6e-1)
    ddpg = recnn.nn.DDPG(policy_net, value_net)
    ost = _oracle_state(policy_net, value_net, 1e-3, 0.0, 1e-2)
    ddpg = ddpg.to(cuda)
    Opt = torch.optim.Adam if mode == "fused" else _PlainAdam
    ddpg.optimizers["value_optimizer"] = Opt(value_net.parameters(), lr=1e-3, weight_decay=1e-2)
    ddpg.optimizers["policy_optimizer"] = Opt(policy_net.parameters(), lr=1e-3)
    ddpg.params["policy_step"] = 3
    ost.params["policy_step"] = 3
    gen = torch.Generator().manual_seed(9)
    n = 0
    for batch in env.train_dataloader:
        B = batch["state"].shape[0]
        masks = [(torch.rand(B, 256, generator=gen) < 0.5).to(torch.uint8) for _ in range(6)]
        users = batch["meta"]["users"].tolist()
        ref_b = O.frame_batch([user_dict[u]["items"] for u in users], [user_dict[u]["ratings"] for u in users], table, 10)
        ref = O.ddpg_step(ost, ref_b, masks, step=ddpg._step, learn=True)
        with fused.external_randomness(ddpg.nets, masks=masks):
            loss = ddpg.update(batch, learn=True)
        ddpg.step()
        assert loss["step"] == n
        assert abs(loss["value"] - ref["value"]) <= 1e-4 * abs(ref["value"]) + 1e-6, (n, loss, ref)
        # the policy loss is a signed mean of Q values of size O(1) that passes through zero: 1e-4 of the Q scale
        assert abs(loss["policy"] - ref["policy"]) <= 1e-4 * max(abs(ref["policy"]), 0.5), (n, loss, ref)
        n += 1
    assert n == len(env.train_dataloader)
    for mod, refp in ((policy_net, ost.policy), (value_net, ost.value), (ddpg.nets["target_policy_net"], ost.target_policy),
                      (ddpg.nets["target_value_net"], ost.target_value)):
        got = O.params_from_module(mod)
        for k in O.PARAM_ORDER:
            assert fro_err(got[k], refp[k]) < 3e-3, k
    # state_dict round trip into a fresh module (streamlit_demo.py:151-161 usage)
    sd = {k: v.cpu() for k, v in policy_net.state_dict().items()}
    fresh = recnn.nn.Actor(1290, 128, 256)
    fresh.load_state_dict(sd)
    assert torch.equal(fresh.linear2.weight, policy_net.linear2.weight.cpu())
    # optimizer state is visible through the torch optimizer object
    st = ddpg.optimizers["value_optimizer"].state[value_net.linear1.weight]
    assert int(st["step"]) == n and st["exp_avg"].abs().sum() > 0
    # a test step: no parameter changes, debug tensors filled
    before = policy_net.linear1.weight.detach().clone()
    lo = ddpg.update(env.test_batch(), learn=False)
    assert torch.equal(before, policy_net.linear1.weight) and "gen_action" in ddpg.debug and lo["value"] > 0


def test_external_weight_changes_are_picked_up(recnn, cuda):
    """load_state_dict / manual edits between updates must reach the compute-layout shadows."""
    from recnn_amd.nn import fused
    fused.set_defaults(mask_mode="none")            # deterministic forward: no dropout noise between the two calls
    ddpg = recnn.nn.DDPG(recnn.nn.Actor(1290, 128, 256, 6e-1), recnn.nn.Critic(1290, 128, 256, 54e-2)).to(cuda)
    batch = _ci_batch(cuda)
    l0 = ddpg.update(batch, learn=False)
    with torch.no_grad():
        ddpg.nets["value_net"].linear3.bias.add_(5.0)
    l1 = ddpg.update(batch, learn=False)
    assert abs((l1["policy"] - l0["policy"]) + 5.0) < 1e-3          # policy loss = -mean(Q): shifts by -5
    sd = {k: torch.zeros_like(v) for k, v in ddpg.nets["policy_net"].state_dict().items()}
    ddpg.nets["policy_net"].load_state_dict(sd)
    ddpg.update(batch, learn=False)
    assert ddpg.debug["gen_action"].abs().max() == 0


def test_install_as_recnn(recnn, cuda):
    recnn.install_as("recnn")
    import recnn as alias
    import recnn.nn as alias_nn
    from recnn.data.env import FrameEnv
    assert alias is recnn and alias_nn.DDPG is recnn.nn.DDPG and FrameEnv is recnn.data.env.FrameEnv


@pytest.mark.parametrize("n_users,policy_step", [(60, 2), (900, 3)])
def test_fused_run_equals_update_loop(recnn, cuda, n_users, policy_step):
    """algo.attach_env(env, rows).run(n) == n x (update(batch); step()) on the same batches, bit for bit.

    900 users = 70+ batches per epoch: the replay then goes through the multi-step run graph (whole policy cycles per
    graph launch, the gather of step t+1 riding on step t's optimizer launch into the second batch buffer set).

    The reference loop is driven with the batches the fused sampler would build: same epoch permutation (taken from the
    fused context), same fixed row count, hash dropout masks keyed by the same device step counter."""
    from recnn_amd.nn import fused
    fused.set_defaults(dtype="bf16", mask_mode="hash", seed=21)
    env, user_dict, table = _env(recnn, cuda, rows_per_batch=96, n_users=n_users, seed=8, n_train=32 if n_users == 60 else 864)
    results = []
    for mode in ("fused", "loop"):
        torch.manual_seed(12)
        ddpg = recnn.nn.DDPG(recnn.nn.Actor(1290, 128, 256, 6e-1), recnn.nn.Critic(1290, 128, 256, 54e-2)).to(cuda)
        ddpg.params["policy_step"] = policy_step
        torch.manual_seed(99)                       # the epoch permutation is drawn from the CPU generator
        ddpg.attach_env(env, rows_per_batch=96, users_per_batch=12)
        ctx = ddpg._fused_ctx
        n = ctx.sampler["n_batches"]
        assert n == len(env.base.train_user_dataset.users) // 12
        if mode == "fused":
            out, hist = ddpg.run(n, history=True)
            assert out["step"] == n - 1 and ddpg._step == n and len(hist) == n
        else:
            hist = []
            perm = ctx.perm.cpu().numpy()
            ids = env.store.user_ids
            ctx.engine.unbind_sampler()
            for i in range(n):
                users = [ids[j] for j in perm[i * 12:(i + 1) * 12]]
                batch = env.collate_users(users)
                assert batch["state"].shape[0] == 96
                out = ddpg.update(batch, learn=True)
                hist.append(dict(out))
                ddpg.step()
        torch.cuda.synchronize()
        histories = locals().get("histories", [])
        histories.append(hist)
        results.append((out["value"], out["policy"], {k: v.detach().clone() for k, v in ddpg.nets["policy_net"].state_dict().items()},
                        int(ddpg.optimizers["value_optimizer"].state[ddpg.nets["value_net"].linear1.weight]["step"])))
    assert results[0][0] == results[1][0] and results[0][1] == results[1][1]
    for k in results[0][2]:
        assert torch.equal(results[0][2][k], results[1][2][k]), k
    assert results[0][3] == results[1][3] == n
    # per-step losses kept on the device during the replay == what update() returned step by step (summation order of
    # the final reductions differs: relative 1e-5)
    for a, b in zip(histories[0], histories[1]):
        assert a["step"] == b["step"]
        for k in ("value", "policy"):
            assert abs(a[k] - b[k]) <= 1e-5 * max(abs(b[k]), 1.0), (a, b)


def test_dp_run_with_sampler_equals_fused_run(recnn, cuda):
    """DataParallelStepper.run (one rank, gloo): merged tail+head phase graphs, two batch buffer sets with the gather of
    step t+1 riding on step t's optimizer launch == the single-GPU run graphs, bit for bit."""
    import os
    import torch.distributed as dist
    from recnn_amd.nn import fused
    from recnn_amd.parallel import DataParallelStepper
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29578")
        dist.init_process_group("gloo", rank=0, world_size=1)
    fused.set_defaults(dtype="bf16", mask_mode="hash", seed=33)
    env, user_dict, table = _env(recnn, cuda, rows_per_batch=96, n_users=900, seed=8, n_train=864)
    results = []
    n = 40
    for mode in ("fused", "dp"):
        torch.manual_seed(12)
        ddpg = recnn.nn.DDPG(recnn.nn.Actor(1290, 128, 256, 6e-1), recnn.nn.Critic(1290, 128, 256, 54e-2)).to(cuda)
        ddpg.params["policy_step"] = 3
        torch.manual_seed(99)
        ddpg.attach_env(env, rows_per_batch=96, users_per_batch=12)
        ctx = ddpg._fused_ctx
        assert ctx.sampler["n_batches"] > n
        if mode == "fused":
            ddpg.run(n)
        else:
            side = torch.cuda.Stream()
            with torch.cuda.stream(side):
                dp = DataParallelStepper(ctx.engine, 96)
                assert ctx.engine.dp_sets() == 2
                dp.run(0, n)
            side.synchronize()
        torch.cuda.synchronize()
        lo = ctx.engine.losses()
        results.append((lo, {k: v.detach().clone() for k, v in ddpg.nets["policy_net"].state_dict().items()},
                        {k: v.detach().clone() for k, v in ddpg.nets["value_net"].state_dict().items()},
                        ctx.engine.loss_history(n)))
    assert results[0][0] == results[1][0], (results[0][0], results[1][0])
    for a, b in zip(results[0][3], results[1][3]):       # every step's losses, not only the last one's
        for k in ("value", "policy"):
            assert abs(a[k] - b[k]) <= 1e-5 * max(abs(b[k]), 1.0), (a, b)
    for j in (1, 2):
        for k in results[0][j]:
            assert torch.equal(results[0][j][k], results[1][j][k]), k


def test_optim_adam_matches_torch_adam(recnn, cuda):
    """recnn_amd.optim.Adam (HIP kernel) against torch.optim.Adam on identical gradients, several steps, weight decay."""
    torch.manual_seed(0)
    w0 = torch.randn(300, 70, device=cuda)
    grads = [torch.randn(300, 70, device=cuda) * 1e-2 for _ in range(5)]
    pa, pb = torch.nn.Parameter(w0.clone()), torch.nn.Parameter(w0.clone())
    oa = recnn.optim.Adam([pa], lr=3e-3, betas=(0.8, 0.99), eps=1e-6, weight_decay=1e-2)
    ob = torch.optim.Adam([pb], lr=3e-3, betas=(0.8, 0.99), eps=1e-6, weight_decay=1e-2)
    for g in grads:
        pa.grad, pb.grad = g.clone(), g.clone()
        oa.step()
        ob.step()
    assert rel_err(pa, pb) < 1e-6
    assert rel_err(oa.state[pa]["exp_avg_sq"], ob.state[pb]["exp_avg_sq"]) < 1e-6 and oa.state[pa]["step"] == 5
    assert recnn.optim.adam_config(oa)["beta2"] == 0.99 and recnn.optim.adam_config(torch.optim.SGD([pb], lr=0.1)) is None


def test_soft_update_utility_matches_reference_formula(recnn, cuda):
    a, b = recnn.nn.Actor(27, 8, 16).to(cuda), recnn.nn.Actor(27, 8, 16).to(cuda)
    before = [p.detach().clone() for p in b.parameters()]
    recnn.utils.soft_update(a, b, soft_tau=0.25)
    for p, t0, t1 in zip(a.parameters(), before, b.parameters()):
        assert torch.allclose(t1, t0 * (1.0 - 0.25) + p * 0.25, rtol=1e-6, atol=1e-7)      # utils/misc.py:3-5


def test_value_update_function(recnn, cuda):
    """recnn.nn.update.value_update: the critic half of the DDPG step, same signature as misc.py:10-20."""
    from recnn_amd.nn import fused
    fused.set_defaults(dtype="fp32", mask_mode="none")
    ddpg = recnn.nn.DDPG(recnn.nn.Actor(1290, 128, 256, 6e-1), recnn.nn.Critic(1290, 128, 256, 54e-2)).to(cuda)
    batch = _ci_batch(cuda)
    batch["done"] = (batch["done"] > 1).float()
    ost = _oracle_state(ddpg.nets["policy_net"], ddpg.nets["value_net"], 1e-5, 1e-2, 1e-2)
    before = ddpg.nets["value_net"].linear1.weight.detach().clone()
    loss = recnn.nn.update.value_update(batch, ddpg.params, ddpg.nets, ddpg.optimizers, device=cuda, learn=True, step=0)
    ref = O.ddpg_step(ost, {k: v.cpu() for k, v in batch.items()}, [], step=1, learn=True)      # step=1: no policy update
    assert abs(float(loss) - ref["value"]) <= 1e-4 * ref["value"]
    assert not torch.equal(before, ddpg.nets["value_net"].linear1.weight)
    assert fro_err(ddpg.nets["value_net"].linear2.weight, ost.value["w2"]) < 1e-3
    assert torch.equal(ddpg.nets["policy_net"].linear1.weight.cpu(), ost.policy["w1"])           # actor untouched


@pytest.mark.parametrize("mode", ["fused", "generic"])
def test_td3_api_matches_oracle(recnn, cuda, mode):
    from recnn_amd.nn import fused
    fused.set_defaults(dtype="fp32", mask_mode="hash", seed=3)
    torch.manual_seed(4)
    v1, v2 = recnn.nn.Critic(1290, 128, 256, 54e-2), recnn.nn.Critic(1290, 128, 256, 54e-2)
    pol = recnn.nn.Actor(1290, 128, 256, 6e-1)
    td3 = recnn.nn.TD3(pol, v1, v2)
    ost = O.TD3State.create(O.params_from_module(pol), O.params_from_module(v1), O.params_from_module(v2),
                            O.AdamState(lr=1e-3), O.AdamState(lr=1e-3), O.AdamState(lr=1e-3))
    td3 = td3.to(cuda)
    Opt = torch.optim.Adam if mode == "fused" else _PlainAdam
    td3.optimizers = {"policy_optimizer": Opt(pol.parameters(), lr=1e-3), "value_optimizer1": Opt(v1.parameters(), lr=1e-3),
                      "value_optimizer2": Opt(v2.parameters(), lr=1e-3)}
    td3.params["policy_update"] = 2
    ost.params["policy_update"] = 2
    gen = torch.Generator().manual_seed(5)
    B = 200
    for t in range(4):
        batch = {"state": torch.randn(B, 1290, generator=gen), "action": torch.randn(B, 128, generator=gen),
                 "reward": torch.randn(B, generator=gen), "next_state": torch.randn(B, 1290, generator=gen),
                 "done": (torch.rand(B, generator=gen) < 0.1).float()}
        masks = [(torch.rand(B, 256, generator=gen) < 0.5).to(torch.uint8) for _ in range(8)]
        noise = torch.randn(B, 128, generator=gen) * 0.5
        ref = O.td3_step(ost, batch, noise, masks, step=t, learn=True)
        with fused.external_randomness(td3.nets, masks=masks, noise=noise, algo="td3"):
            lo = td3.update({k: v.to(cuda) for k, v in batch.items()}, learn=True)
        td3.step()
        for k in ("value1", "value2"):
            assert abs(lo[k] - ref[k]) <= 1e-4 * ref[k], (t, k, lo, ref)
        assert abs(lo["policy"] - ref["policy"]) <= 1e-4 * max(abs(ref["policy"]), 0.5), (t, lo, ref)
    # td3.py:136-141: target policy never soft-updated; target critics are
    assert torch.equal(td3.nets["target_policy_net"].linear1.weight.cpu(), ost.target_policy["w1"])
    assert fro_err(td3.nets["target_value_net2"].linear1.weight, ost.target_value2["w1"]) < 1e-3
    assert fro_err(pol.linear3.weight, ost.policy["w3"]) < 3e-3
    # learn=False: losses only, debug tensors, nothing changes
    w = v1.linear1.weight.detach().clone()
    lo = td3.update({k: v.to(cuda) for k, v in batch.items()}, learn=False)
    assert torch.equal(w, v1.linear1.weight) and td3.debug["next_action"].shape == (B, 128) and lo["value2"] > 0


def test_frame_env_custom_embed_batch(recnn, cuda):
    """A user-supplied embed_batch callable receives the windowed index batch, as in the reference (utils.py:181-187)."""
    seen = {}

    def my_embed(batch, item_embeddings_tensor, frame_size, *a, **k):
        seen["items"] = batch["items"]
        out = recnn.data.batch_tensor_embeddings(batch, item_embeddings_tensor, frame_size)
        out["custom"] = True
        return out
    items, ratings, table = make_store(n_users=12, n_items=100, emb_dim=128, min_len=11, max_len=30, seed=6)
    user_dict = {u: {"items": items[u], "ratings": ratings[u]} for u in range(12)}
    env = recnn.data.env.FrameEnv.from_user_dict(torch.from_numpy(table), user_dict, list(range(10)), [10, 11], frame_size=10,
                                                 batch_size=4, device=cuda, embed_batch=my_embed)
    b = env.train_batch()
    users = b["meta"]["users"].tolist()
    ref = O.frame_batch([items[u] for u in users], [ratings[u] for u in users], table, 10)
    assert b["custom"] and seen["items"].shape == (ref["state"].shape[0], 11)
    assert np.array_equal(seen["items"].cpu().numpy(), ref["items"])
    for k in ("state", "action", "reward", "next_state", "done"):
        assert np.array_equal(b[k].cpu().numpy(), ref[k]), k
