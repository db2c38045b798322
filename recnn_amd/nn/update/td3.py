"""td3_update (reference: recnn/nn/update/td3.py:8-150) on the fused HIP step engine.

Quirks kept: target action = target_policy(next_state) + clamp(N(0, noise_std), +-noise_clip); twin target
critics, min, NO clamp of the TD target; both critics updated every step (MSELoss); policy loss through
value_net1 every step; on `step % policy_update == 0` the actor update (with the L1 clip quirk) and soft updates
of BOTH target critics -- the target policy net is never soft-updated (td3.py:136-141).
"""
import torch

from ... import _lib as L
from ... import utils
from .. import fused

__all__ = ["td3_update"]


def td3_update(batch, params, nets, optimizer, device=torch.device("cpu"), debug=None, writer=utils.DummyWriter(),
               learn=False, step=-1):
    """
    :param params: dict(gamma, noise_std, noise_clip, soft_tau, policy_update)
    :param nets: dict(value_net1, target_value_net1, value_net2, target_value_net2, policy_net, target_policy_net)
    :param optimizer: dict(policy_optimizer, value_optimizer1, value_optimizer2)
    :return: {"value1": float, "value2": float, "policy": float, "step": step}
    """
    if debug is None:
        debug = dict()
    ctx = fused.context_for("td3", nets)
    ctx.ensure(nets, batch["state"].shape[0])
    rows = ctx.load_batch(batch)
    eng = ctx.engine
    keys = ("policy_optimizer", "value_optimizer1", "value_optimizer2")
    cfgs = fused.fused_adam_configs(optimizer, keys) if learn else None
    if cfgs and cfgs[1] != cfgs[2]:
        cfgs = None                          # the engine shares one Adam configuration between the twin critics
    ctx.set_hyper(params, cfgs[0] if cfgs else None, cfgs[1] if cfgs else None)
    ctx.apply_external(rows)
    policy_step = bool(learn) and (step % params["policy_update"] == 0)
    s = L.current_stream()
    if not learn or cfgs:
        if learn:
            ctx.mirror_optimizer_state(optimizer["policy_optimizer"], L.NET_POLICY)
            ctx.mirror_optimizer_state(optimizer["value_optimizer1"], L.NET_VALUE1)
            ctx.mirror_optimizer_state(optimizer["value_optimizer2"], L.NET_VALUE2)
        L.call("recnn_engine_step", eng.handle, rows, int(bool(learn)), int(step), s)
        if learn:
            ctx.bump(optimizer["value_optimizer1"], L.NET_VALUE1)
            ctx.bump(optimizer["value_optimizer2"], L.NET_VALUE2)
            if policy_step:
                ctx.bump(optimizer["policy_optimizer"], L.NET_POLICY)
            ctx.mark_stepped((L.NET_VALUE1, L.NET_VALUE2)
                             + ((L.NET_POLICY, L.NET_TARGET_VALUE1, L.NET_TARGET_VALUE2) if policy_step else ()))
    else:
        L.call("recnn_engine_value_grads", eng.handle, rows, 1, s)
        ctx.attach_grads(L.NET_VALUE1)
        ctx.attach_grads(L.NET_VALUE2)
        optimizer["value_optimizer1"].step()
        optimizer["value_optimizer2"].step()
        eng.refresh(L.NET_VALUE1)
        eng.refresh(L.NET_VALUE2)
        L.call("recnn_engine_policy_grads", eng.handle, rows, int(policy_step), s)
        if policy_step:
            L.call("recnn_engine_clip_policy_grads", eng.handle, 1.0, s)
            ctx.attach_grads(L.NET_POLICY)
            optimizer["policy_optimizer"].step()
            eng.refresh(L.NET_POLICY)
            tau = float(params["soft_tau"])
            L.call("recnn_engine_soft_update", eng.handle, L.NET_VALUE1, L.NET_TARGET_VALUE1, tau, s)
            L.call("recnn_engine_soft_update", eng.handle, L.NET_VALUE2, L.NET_TARGET_VALUE2, tau, s)
        L.call("recnn_engine_finish", eng.handle, rows, 0, 0, s)
        ctx._sync_versions()
    if not learn:
        debug["next_action"] = eng.buffer("next_action", rows)
        debug["gen_action"] = eng.buffer("gen_action", rows)
        if not isinstance(writer, utils.DummyWriter):
            writer.add_figure("next_action", utils.pairwise_distances_fig(debug["next_action"][:50]), step)
            writer.add_histogram("value1", eng.buffer("q1", rows), step)
            writer.add_histogram("value2", eng.buffer("q2", rows), step)
            writer.add_histogram("target_value", eng.buffer("target_q", rows), step)
            writer.add_histogram("expected_value", eng.buffer("expected", rows), step)
            writer.add_figure("gen_action", utils.pairwise_distances_fig(debug["gen_action"][:50]), step)
            writer.add_histogram("policy_loss", -eng.buffer("q_pi", rows), step)
    lo = eng.losses()
    losses = {"value1": lo["value1"], "value2": lo["value2"], "policy": lo["policy"], "step": step}
    utils.write_losses(writer, losses, kind="train" if learn else "test")
    return losses
