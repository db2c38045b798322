"""temporal_difference / value_update (reference: recnn/nn/update/misc.py:6-55).

`value_update` keeps the reference signature.  It is the critic half of the DDPG step: TD target from the target
nets, critic forward/backward, critic optimizer step -- executed by the fused HIP engine.  It returns the value
loss as a 0-dim tensor like the reference.
"""
import torch

from ... import _lib as L
from ... import utils
from .. import fused

__all__ = ["temporal_difference", "value_update"]


def temporal_difference(reward, done, gamma, target):
    """reward + (1 - done) * gamma * target (misc.py:6-7).  Elementwise helper for user code; inside the step this
    is fused into the critic-head kernel (csrc/head.hip)."""
    return reward + (1.0 - done) * gamma * target


def _log_value_debug(ctx, rows, debug, writer, step):
    eng = ctx.engine
    if debug is not None:
        debug["next_action"] = eng.buffer("next_action", rows)
    if not isinstance(writer, utils.DummyWriter):
        writer.add_figure("next_action", utils.pairwise_distances_fig(eng.buffer("next_action", min(rows, 50))), step)
        writer.add_histogram("value", eng.buffer("q1", rows), step)
        writer.add_histogram("target_value", eng.buffer("target_q", rows), step)
        writer.add_histogram("expected_value", eng.buffer("expected", rows), step)


def _value_update_modules(batch, params, nets, optimizer, device, debug, writer, learn, step):
    """The same critic update for networks the fused DDPG engine does not adopt (the REINFORCE pair: DiscreteActor +
    a Critic over [state | action distribution], misc.py:10-55): module forwards / backward on the HIP GEMM and
    policy-head kernels through autograd, TD target and loss on per-row vectors."""
    from ... import data
    # the networks' device, not get_base_batch's default "cuda" (= cuda:0): misc.py:25 passes device=device
    state, action, reward, next_state, done = data.get_base_batch(batch, device=next(nets["value_net"].parameters()).device)
    with torch.no_grad():
        next_action = nets["target_policy_net"](next_state)
        target_value = nets["target_value_net"](next_state, next_action)
        expected_value = temporal_difference(reward, done, params["gamma"], target_value)
        expected_value = torch.clamp(expected_value, params["min_value"], params["max_value"])
    value = nets["value_net"](state, action)
    value_loss = torch.pow(value - expected_value, 2).mean()
    if learn:
        optimizer["value_optimizer"].zero_grad()
        value_loss.backward()
        optimizer["value_optimizer"].step()
    else:
        if debug is not None:
            debug["next_action"] = next_action
        if not isinstance(writer, utils.DummyWriter):
            writer.add_figure("next_action", utils.pairwise_distances_fig(next_action[:50]), step)
            writer.add_histogram("value", value, step)
            writer.add_histogram("target_value", target_value, step)
            writer.add_histogram("expected_value", expected_value, step)
    return value_loss


def value_update(batch, params, nets, optimizer, device=torch.device("cpu"), debug=None, writer=utils.DummyWriter(),
                 learn=False, step=-1):
    from ..models import Actor
    # the fused engine implements exactly Actor's forward: a subclass with another forward goes through the modules
    if type(nets["target_policy_net"]) is not Actor:
        return _value_update_modules(batch, params, nets, optimizer, device, debug, writer, learn, step)
    ctx = fused.context_for("ddpg", nets)
    rows = batch["state"].shape[0]
    ctx.ensure(nets, rows)
    rows = ctx.load_batch(batch)
    eng = ctx.engine
    cfg = fused.fused_adam_configs(optimizer, ("value_optimizer",)) if learn else None
    ctx.set_hyper(params, None, cfg[0] if cfg else None)
    ctx.apply_external(rows)
    s = L.current_stream()
    L.call("recnn_engine_value_grads", eng.handle, rows, int(learn), s)
    if learn:
        opt = optimizer["value_optimizer"]
        if cfg:
            ctx.mirror_optimizer_state(opt, L.NET_VALUE1)
            L.call("recnn_engine_value_apply", eng.handle, 0, 1.0, s)
            ctx.bump(opt, L.NET_VALUE1)
            ctx.mark_stepped((L.NET_VALUE1,))
        else:
            ctx.attach_grads(L.NET_VALUE1)
            opt.step()
            eng.refresh(L.NET_VALUE1)
            ctx._sync_versions()
    else:
        _log_value_debug(ctx, rows, debug, writer, step)
    # close the step on the device: ticks the mask-key step counter and, for the fused optimizer, the critic's Adam
    # step counter (apply_net reads t = *t_ptr + 1) -- without it repeated calls reuse the same dropout masks and
    # apply the bias correction of step 1 forever
    L.call("recnn_engine_finish", eng.handle, rows, int(bool(learn and cfg)), 0, s)
    q, y = eng.buffer("q1", rows), eng.buffer("expected", rows)
    return torch.pow(q - y, 2).mean()
