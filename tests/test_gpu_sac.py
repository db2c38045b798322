"""recnn_amd.nn.soft_q_update + StateCritic / SoftQ / StochasticActor on the HIP kernels against the golden runs of the
reference notebook's own SAC cells (tests/golden/sac_*.npz): fed the run's batches, z draws and dropout masks, the step
reproduces its losses and all four networks (fp32, 1e-4 relative)."""
import os

import numpy as np
import pytest
import torch

from tests import sac_replay as SR
from tests.helpers import rel_err

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _run(name, optimizer):
    import recnn_amd
    from oracle import sac_oracle as SO
    fx = SR.load(os.path.join(GOLDEN, name + ".npz"))
    g = fx["g"]
    dev = torch.device("cuda")
    S, A, H = fx["S"], fx["A"], fx["H"]
    value, target = recnn_amd.nn.StateCritic(S, H), recnn_amd.nn.StateCritic(S, H)
    soft_q = recnn_amd.nn.SoftQ(S, A, H)
    policy = recnn_amd.nn.StochasticActor(S, A, H, SR.ACTOR)
    with torch.no_grad():
        for mod, tag, snap in ((value, "value", SO.critic_params_from_module), (target, "value", SO.critic_params_from_module),
                               (soft_q, "soft_q", SO.critic_params_from_module), (policy, "policy", SO.policy_params_from_module)):
            for k, p in snap(mod).items():
                p.copy_(torch.from_numpy(g[f"{tag}.{k}"]))
    nets = {"value_net": value.to(dev), "target_value_net": target.to(dev), "soft_q_net": soft_q.to(dev), "policy_net": policy.to(dev)}
    opt = {"value_optimizer": optimizer(value.parameters(), lr=fx["lr"], weight_decay=fx["wd"]),
           "soft_q_optimizer": optimizer(soft_q.parameters(), lr=fx["lr"], weight_decay=fx["wd"]),
           "policy_optimizer": optimizer(policy.parameters(), lr=fx["lr"], weight_decay=fx["wd"])}
    bs = SR.batches(fx, dev)
    policy.forced_masks = []
    losses = []
    for t in range(fx["steps"]):
        m = [torch.from_numpy(x).to(dev) for x in g["masks"][t]]
        policy.forced_masks[:] = [(m[0], m[1])]
        policy.forced_z[:] = [float(g["z"][t])]
        out = recnn_amd.nn.soft_q_update(bs[t % 2], SR.PARAMS, nets, opt, learn=True, step=t)
        assert not policy.forced_masks and not policy.forced_z
        losses.append([t, out["value"], out["softq"], out["policy"]])
    return fx, np.asarray(losses), nets


@pytest.mark.parametrize("name", ["sac_small", "sac_wd"])
def test_soft_q_update_replays_the_notebook_run(cuda, name):
    from oracle import sac_oracle as SO
    fx, losses, nets = _run(name, torch.optim.Adam)
    assert rel_err(losses[:, 1:], fx["g"]["losses"][:, 1:]) < 1e-4, (losses, fx["g"]["losses"])
    for tag, key, snap in (("value", "value_net", SO.critic_params_from_module), ("target_value", "target_value_net", SO.critic_params_from_module),
                           ("soft_q", "soft_q_net", SO.critic_params_from_module), ("policy", "policy_net", SO.policy_params_from_module)):
        for k, v in snap(nets[key]).items():
            assert rel_err(v, fx["g"][f"final.{tag}.{k}"]) < 1e-4, (tag, k)


def test_soft_q_update_with_the_hip_adam(cuda):
    import recnn_amd
    fx, losses, _ = _run("sac_small", recnn_amd.optim.Adam)
    assert rel_err(losses[:, 1:], fx["g"]["losses"][:, 1:]) < 1e-4


def test_soft_q_update_at_the_notebook_shape_and_eval_mode(cuda):
    """StateCritic(1290, 256), SoftQ(1290, 128, 256), StochasticActor(1290, 128, 256) on a 2048-row batch (SAC.ipynb cell 10): ten
    learning steps run, the losses are finite, every network moves; learn=False changes nothing and fills the debug dict."""
    import recnn_amd
    dev = torch.device("cuda")
    torch.manual_seed(3)
    nets = {"value_net": recnn_amd.nn.StateCritic(1290, 256, 1e-1).to(dev), "target_value_net": recnn_amd.nn.StateCritic(1290, 256).to(dev),
            "soft_q_net": recnn_amd.nn.SoftQ(1290, 128, 256, 2e-1).to(dev), "policy_net": recnn_amd.nn.StochasticActor(1290, 128, 256, SR.ACTOR).to(dev)}
    recnn_amd.utils.soft_update(nets["value_net"], nets["target_value_net"], soft_tau=1.0)
    opt = {k + "_optimizer": recnn_amd.optim.Adam(nets[n].parameters(), lr=1e-4)
           for k, n in (("value", "value_net"), ("soft_q", "soft_q_net"), ("policy", "policy_net"))}
    B = 2048
    batch = {"state": torch.randn(B, 1290, device=dev), "action": torch.randn(B, 128, device=dev) * 0.3, "reward": torch.randn(B, device=dev),
             "next_state": torch.randn(B, 1290, device=dev), "done": (torch.rand(B, device=dev) < 0.1).float()}
    before = {n: [p.detach().clone() for p in m.parameters()] for n, m in nets.items()}
    for t in range(10):
        out = recnn_amd.nn.soft_q_update(batch, SR.PARAMS, nets, opt, learn=True, step=t)
        assert all(np.isfinite(out[k]) for k in ("value", "softq", "policy")) and out["step"] == t
    for n, m in nets.items():
        assert any(not torch.equal(a, b) for a, b in zip(before[n], m.parameters())), n
    snap = {n: [p.detach().clone() for p in m.parameters()] for n, m in nets.items()}
    debug = {}
    nets["policy_net"].eval()
    out = recnn_amd.nn.soft_q_update(batch, SR.PARAMS, nets, opt, debug=debug, learn=False, step=10)
    assert "test next_action" in debug and debug["test next_action"].shape == (B, 128)
    for n, m in nets.items():
        assert all(torch.equal(a, b) for a, b in zip(snap[n], m.parameters())), n


def test_graphed_soft_q_update_equals_the_eager_update(cuda):
    """GraphedUpdate(soft_q_update) at the notebook's shape: the captured step replayed == the step issued eagerly through the same
    device counters, bit for bit (all four networks, losses) over 8 steps."""
    import time
    import recnn_amd
    from recnn_amd.nn import GraphedUpdate, soft_q_update
    dev = torch.device("cuda")
    B, steps = 1024, 10
    gcpu = torch.Generator().manual_seed(4)
    batches = [{"state": torch.randn(B, 1290, generator=gcpu).to(dev), "action": (torch.randn(B, 128, generator=gcpu) * 0.3).to(dev),
                "reward": torch.randn(B, generator=gcpu).to(dev), "next_state": torch.randn(B, 1290, generator=gcpu).to(dev),
                "done": (torch.rand(B, generator=gcpu) < 0.1).float().to(dev)} for _ in range(steps)]
    zs = torch.randn(steps, generator=gcpu)

    def build(graphs):
        torch.manual_seed(3)
        nets = {"value_net": recnn_amd.nn.StateCritic(1290, 256, 1e-1).to(dev), "target_value_net": recnn_amd.nn.StateCritic(1290, 256).to(dev),
                "soft_q_net": recnn_amd.nn.SoftQ(1290, 128, 256, 2e-1).to(dev), "policy_net": recnn_amd.nn.StochasticActor(1290, 128, 256, SR.ACTOR).to(dev)}
        recnn_amd.utils.soft_update(nets["value_net"], nets["target_value_net"], soft_tau=1.0)
        opt = {k + "_optimizer": recnn_amd.optim.Adam(nets[n].parameters(), lr=1e-4, capturable=True)
               for k, n in (("value", "value_net"), ("soft_q", "soft_q_net"), ("policy", "policy_net"))}
        policy = nets["policy_net"]
        z_static = torch.zeros((), device=dev)
        policy.forced_z = [zs[0].to(dev), zs[1].to(dev)]            # the two warm-up steps inside the constructor
        gu = GraphedUpdate(soft_q_update, batches[0], SR.PARAMS, nets, opt, warmup=2, graphs=graphs)
        losses, t_host = [], 0.0
        for t in range(2, steps):
            z_static.copy_(zs[t])
            policy.forced_z[:] = [z_static]                        # a replay reads the tensor its capture consumed
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = gu(batches[t])
            t_host += time.perf_counter() - t0
            losses.append([float(out[k]) for k in ("value", "softq", "policy")])
        torch.cuda.synchronize()
        return {n: [p.detach().clone() for p in m.parameters()] for n, m in nets.items()}, losses, t_host / (steps - 2)

    eager, eager_losses, t_eager = build(False)
    graphed, graphed_losses, t_graph = build(True)
    assert eager_losses == graphed_losses, (eager_losses, graphed_losses)
    for n in eager:
        assert all(torch.equal(a, b) for a, b in zip(eager[n], graphed[n])), n
    print(f"sac step, host time until the call returns: eager {t_eager * 1e3:.2f} ms, graphed {t_graph * 1e3:.2f} ms (incl. the capture)")
