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


def value_update(batch, params, nets, optimizer, device=torch.device("cpu"), debug=None, writer=utils.DummyWriter(),
                 learn=False, step=-1):
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
