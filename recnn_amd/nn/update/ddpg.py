"""ddpg_update (reference: recnn/nn/update/ddpg.py:8-104) on the fused HIP step engine.

Same signature, same in-place effects on `nets` / `optimizer` / `debug`, same returned dict of python floats.
What runs underneath: one call into librecnn_hip.so per update (15 kernel launches, 25 on a policy step)
instead of ~560 ATen calls.  Reference quirks kept: the critic is updated BEFORE the policy loss uses it
(ddpg.py:63-79), the policy loss is computed on every step, `step % policy_step == 0` gates the actor update
(step 0 is a policy step), `clip_grad_norm_(.., -1, 1)` L1-normalises and sign-flips the actor gradient
(ddpg.py:92), soft updates follow the actor step.
"""
import torch

from ... import _lib as L
from ... import utils
from .. import fused
from .misc import _log_value_debug

__all__ = ["ddpg_update"]


def ddpg_update(batch, params, nets, optimizer, device=torch.device("cpu"), debug=None, writer=utils.DummyWriter(),
                learn=False, step=-1):
    """
    :param batch: dict with state, action, reward, next_state, done (FrameEnv batch or any GPU tensors).
    :param params: dict(gamma, min_value, max_value, policy_step, soft_tau)
    :param nets: dict(value_net, target_value_net, policy_net, target_policy_net) of Critic / Actor modules on the GPU.
    :param optimizer: dict(policy_optimizer, value_optimizer).  (recnn_amd|torch).optim.Adam run fused inside the
        engine; any other torch optimizer is stepped by torch between the engine's gradient phases.
    :param device: kept for signature compatibility; the networks' device is what counts (must be a GPU).
    :param debug: dict that receives next_action / gen_action on learn=False.
    :param writer: tensorboard SummaryWriter-like object.
    :param learn: False = test step (losses only).
    :param step: integer step, gates the delayed policy update.
    :return: {"value": float, "policy": float, "step": step}
    """
    ctx = fused.context_for("ddpg", nets)
    ctx.ensure(nets, batch["state"].shape[0])
    rows = ctx.load_batch(batch)
    eng = ctx.engine
    cfgs = fused.fused_adam_configs(optimizer, ("policy_optimizer", "value_optimizer")) if learn else None
    ctx.set_hyper(params, cfgs[0] if cfgs else None, cfgs[1] if cfgs else None)
    ctx.apply_external(rows)
    policy_step = bool(learn) and (step % params["policy_step"] == 0)
    s = L.current_stream()
    if not learn or cfgs:
        if learn:
            ctx.mirror_optimizer_state(optimizer["policy_optimizer"], L.NET_POLICY)
            ctx.mirror_optimizer_state(optimizer["value_optimizer"], L.NET_VALUE1)
        L.call("recnn_engine_step", eng.handle, rows, int(bool(learn)), int(step), s)
        if learn:
            ctx.bump(optimizer["value_optimizer"], L.NET_VALUE1)
            if policy_step:
                ctx.bump(optimizer["policy_optimizer"], L.NET_POLICY)
            ctx.mark_stepped((L.NET_VALUE1,) + ((L.NET_POLICY, L.NET_TARGET_POLICY, L.NET_TARGET_VALUE1) if policy_step else ()))
    else:
        # arbitrary torch optimizers: the engine produces gradients, torch applies them
        L.call("recnn_engine_value_grads", eng.handle, rows, 1, s)
        ctx.attach_grads(L.NET_VALUE1)
        optimizer["value_optimizer"].step()
        eng.refresh(L.NET_VALUE1)
        L.call("recnn_engine_policy_grads", eng.handle, rows, int(policy_step), s)
        if policy_step:
            L.call("recnn_engine_clip_policy_grads", eng.handle, 1.0, s)
            ctx.attach_grads(L.NET_POLICY)
            optimizer["policy_optimizer"].step()
            eng.refresh(L.NET_POLICY)
            L.call("recnn_engine_soft_update", eng.handle, L.NET_VALUE1, L.NET_TARGET_VALUE1, float(params["soft_tau"]), s)
            L.call("recnn_engine_soft_update", eng.handle, L.NET_POLICY, L.NET_TARGET_POLICY, float(params["soft_tau"]), s)
        L.call("recnn_engine_finish", eng.handle, rows, 0, 0, s)
        ctx._sync_versions()
    if not learn:
        _log_value_debug(ctx, rows, debug, writer, step)
        if debug is not None:
            debug["gen_action"] = eng.buffer("gen_action", rows)
        if not isinstance(writer, utils.DummyWriter):
            writer.add_histogram("policy_loss", -eng.buffer("q_pi", rows), step)
            writer.add_figure("next_action", utils.pairwise_distances_fig(eng.buffer("gen_action", min(rows, 50))), step)
    lo = eng.losses()                       # device sync, as the reference's .item() calls
    losses = {"value": lo["value"], "policy": lo["policy"], "step": step}
    utils.write_losses(writer, losses, kind="train" if learn else "test")
    return losses
