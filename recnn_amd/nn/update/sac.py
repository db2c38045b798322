"""soft_q_update -- one Soft Actor-Critic step (reference: `examples/1. Vanilla RL/4. SAC.ipynb`, code cell 8 `soft_q_update`;
SURVEY.md 8 row f4).  The notebook keeps networks, optimizers and criteria in globals and takes `(step, batch, params, learn)`;
here they travel the way the library's update functions take them (recnn/nn/update/ddpg.py:10-21): `nets` = {value_net,
target_value_net, soft_q_net, policy_net}, `optimizer` = {value_optimizer, soft_q_optimizer, policy_optimizer}.

Every GEMM (the three critics' stacks, the policy trunk and its two heads, forward and backward) runs on csrc/gemm.hip
through `recnn_amd.nn.functional`; the soft update on csrc/optim.hip; torch is left with elementwise algebra on [B, 1] and
[B, action_dim] tensors (the squashed-Gaussian log-prob, the three MSE / policy losses) and autograd's bookkeeping.

Kept from the notebook, quirks included:
  * the policy is evaluated on `state` (the result is still called next_action);
  * `next_value = Q(state, next_action) - log_prob` broadcasts [B, 1] - [B, action_dim], and the value loss is the MSE of the
    [B, 1] prediction against that [B, action_dim] target (torch warns, then broadcasts);
  * Q(state, next_action) is evaluated AFTER the soft-Q optimizer step (with the updated weights), V(state) before its own;
  * the policy gradient flows through log_prob only (the advantage-like factor is detached); the regularisers
    mean_lambda * mean^2, std_lambda * log_std^2, z_lambda * z^2 are added as written (z is one scalar per call);
  * only the state-value target network exists and is soft-updated, right after the value step.
"""
import torch

from ... import data, utils

__all__ = ["soft_q_update"]


def soft_q_update(batch, params, nets, optimizer, device=torch.device("cpu"), debug=None, writer=utils.DummyWriter(),
                  learn=True, step=-1):
    """
    :param batch: batch [state, action, reward, next_state, done] returned by environment.
    :param params: dict(gamma, soft_tau, mean_lambda, std_lambda, z_lambda)
    :param nets: dict(value_net, target_value_net, soft_q_net, policy_net)
    :param optimizer: dict(value_optimizer, soft_q_optimizer, policy_optimizer)
    :param device: accepted for signature compatibility; the step runs where the networks live (the GPU)
    :param debug: dictionary where debug data about actions is saved
    :param writer: torch.SummaryWriter
    :param learn: whether to learn on this step (used for testing)
    :param step: integer step for the loss dictionary / writer
    :return: loss dictionary {"value", "softq", "policy", "step"}
    """
    if debug is None:
        debug = dict()
    dev = next(nets["policy_net"].parameters()).device
    state, action, reward, next_state, done = data.get_base_batch(batch, device=dev)
    mse = torch.nn.functional.mse_loss

    # ---- soft Q ------------------------------------------------------------------------------------------------------
    expected_softq_value = nets["soft_q_net"](state, action)
    expected_value = nets["value_net"](state)
    next_action, log_prob, z, mean, log_std = nets["policy_net"].evaluate(state)
    with torch.no_grad():
        target_value = nets["target_value_net"](next_state)
        next_q_value = reward + (1 - done) * params["gamma"] * target_value
    q_value_loss = mse(expected_softq_value, next_q_value)
    if learn:
        optimizer["soft_q_optimizer"].zero_grad()
        q_value_loss.backward()
        optimizer["soft_q_optimizer"].step()

    # ---- state value -------------------------------------------------------------------------------------------------
    with torch.no_grad():
        expected_next_softq_value = nets["soft_q_net"](state, next_action.detach())
        next_value = expected_next_softq_value - log_prob.detach()           # [B, 1] - [B, action_dim]
    value_loss = ((expected_value - next_value) ** 2).mean()                 # = nn.MSELoss() with its broadcast
    if learn:
        optimizer["value_optimizer"].zero_grad()
        value_loss.backward()
        optimizer["value_optimizer"].step()
        utils.soft_update(nets["value_net"], nets["target_value_net"], soft_tau=params["soft_tau"])

    # ---- policy ------------------------------------------------------------------------------------------------------
    log_prob_target = (expected_next_softq_value - expected_value).detach()
    policy_loss = (log_prob * (log_prob - log_prob_target).detach()).mean()
    mean_loss = params["mean_lambda"] * mean.pow(2).mean()
    std_loss = params["std_lambda"] * log_std.pow(2).mean()
    z_loss = params["z_lambda"] * z.pow(2).sum(0).mean()
    policy_loss = policy_loss + mean_loss + std_loss + z_loss
    if learn:
        debug["next_action"] = next_action
        optimizer["policy_optimizer"].zero_grad()
        policy_loss.backward()
        optimizer["policy_optimizer"].step()
    else:
        debug["test next_action"] = next_action
        if not isinstance(writer, utils.DummyWriter):
            writer.add_figure("next_action", utils.pairwise_distances_fig(next_action[:50]), step)
            for name, t in (("expected_softq_value", expected_softq_value), ("expected_value", expected_value), ("log_prob", log_prob),
                            ("z", z), ("mean", mean), ("log_std", log_std), ("target_value", target_value),
                            ("next_q_value", next_q_value), ("expected_next_softq_value", expected_next_softq_value),
                            ("next_value", next_value), ("log_prob_target", log_prob_target), ("mean_loss", mean_loss),
                            ("std_loss", std_loss), ("z_loss", z_loss)):
                writer.add_histogram(name, t, step)
        writer.close()

    # (inside a stream capture -- recnn_amd.nn.graphed.GraphedUpdate -- the losses stay device scalars: .item() would sync)
    val = (lambda t: t.detach()) if (dev.type == "cuda" and torch.cuda.is_current_stream_capturing()) else (lambda t: t.item())
    losses = {"value": val(value_loss), "softq": val(q_value_loss), "policy": val(policy_loss), "step": step}
    return losses
