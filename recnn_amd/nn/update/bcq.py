"""bcq_update (reference: recnn/nn/update/bcq.py:11-179) -- SURVEY.md 8 row f4.

One Batch-Constrained Q-learning step: (1) the conditional VAE (`generator_net`) learns to reconstruct the logged action,
(2) the critic regresses onto the TD target built from the BEST of `n_generator_samples` perturbed candidate actions per
next state, (3) every `perturbator_step` steps the perturbator climbs the critic, then the three soft target updates.

Every GEMM of the step (VAE encoder / decoder stacks, perturbator, critics, forward and backward) runs on gemm.hip through
`recnn_amd.nn.functional`, the VAE's latent layer and loss on csrc/vae.hip, soft updates on optim.hip; torch is left with
algebra on per-row vectors ([B, 1] TD targets, the max over candidates) and autograd's bookkeeping.

Kept from the reference, quirks included:
  * `target_value_net1` provides BOTH target Q values (bcq.py:105-106), so 0.75 min + 0.25 max is 0.75 q + 0.25 q of one
    critic.  With the target in eval mode the second forward is bit-identical to the first and is not repeated; a target
    left in train mode (fresh dropout per call) is evaluated twice like the reference does;
  * `value_net2` gets no gradient (its optimizer's step is a no-op) but `target_value_net2` still tracks it;
  * the perturbator loss is evaluated and returned on every call; backward + `clip_grad_norm_(.., -1, 1)` (the L1-normalise
    and sign-flip quirk) + optimizer step only on `step % perturbator_step == 0`; soft updates on every learning call;
  * the optimizer key is `perturbator_optimizer` (the reference's docstring says `policy_optimizer`, its code does not).
One deliberate difference: the reference imports `torch.functional as F` (bcq.py:2), which has no `mse_loss`, and raises
AttributeError as written; this function computes the loss the line spells (`F.mse_loss` of `torch.nn.functional`).
Not reproduced: gradients the reference leaves in `.grad` of networks that are not being stepped (the critic's and the
VAE decoder's from the perturbator loss) -- every optimizer in the step zeroes its gradients before use, so parameters,
optimizer state and losses are unaffected, and the GEMMs that would compute them are skipped.
"""
import torch

from ... import data, utils
from .. import functional as F_hip
from ..models import Critic, bcqGenerator, bcqPerturbator
from .misc import temporal_difference

__all__ = ["bcq_update"]


def _score(critic, state, action):
    """critic(state, action) with the critic's weights held constant (gradient w.r.t. the action only)."""
    if type(critic) is Critic and state.is_cuda:
        return F_hip.mlp_frozen(torch.cat([state, action], 1), critic, critic.training)
    return critic(state, action)


def bcq_update(batch, params, nets, optimizer, device=torch.device("cpu"), debug=None, writer=utils.DummyWriter(),
               learn=False, step=-1):
    """
    :param batch: batch [state, action, reward, next_state] returned by environment.
    :param params: dict(gamma, soft_tau, n_generator_samples, perturbator_step)
    :param nets: dict(generator_net, perturbator_net, target_perturbator_net, value_net1, target_value_net1, value_net2,
                 target_value_net2)
    :param optimizer: dict(generator_optimizer, perturbator_optimizer, value_optimizer1, value_optimizer2)
    :param device: accepted for signature compatibility; the step runs where the networks live (the GPU)
    :param debug: dictionary where debug data about actions is saved
    :param writer: torch.SummaryWriter
    :param learn: whether to learn on this step (used for testing)
    :param step: integer step for the perturbator update
    :return: loss dictionary {"value", "perturbator", "generator", "step"}
    """
    if debug is None:
        debug = dict()
    generator = nets["generator_net"]
    dev = next(generator.parameters()).device
    state, action, reward, next_state, done = data.get_base_batch(batch, device=dev)
    batch_size = done.size(0)
    log = not isinstance(writer, utils.DummyWriter)

    # ---- variational auto-encoder ------------------------------------------------------------------------------------
    recon, mean, std = generator(state, action)
    generator_loss = F_hip.vae_loss(recon, action, mean, std, 0.5)[2]
    if not learn:
        debug["recon"] = recon
        if log:
            writer.add_histogram("generator_mean", mean, step)
            writer.add_histogram("generator_std", std, step)
            writer.add_figure("reconstructed", utils.pairwise_distances_fig(recon[:50]), step)
    if learn:
        optimizer["generator_optimizer"].zero_grad()
        generator_loss.backward()
        optimizer["generator_optimizer"].step()

    # ---- critic: TD target from the best of n perturbed candidates per next state -------------------------------------
    with torch.no_grad():
        n = params["n_generator_samples"]
        tpert, tvalue = nets["target_perturbator_net"], nets["target_value_net1"]
        if (type(generator) is bcqGenerator and type(tpert) is bcqPerturbator and type(tvalue) is Critic and next_state.is_cuda
                and not tpert.training and not tvalue.training):
            # the three networks read [state | candidate]: the state part of each layer 1 is computed once per state row,
            # the repeated states are never materialised (functional.mlp_candidates)
            sampled_action = generator.decode_candidates(next_state, n)
            perturbed_action = tpert.candidates(next_state, sampled_action, n)
            target_Q1 = target_Q2 = tvalue.candidates(next_state, perturbed_action, n)
        else:
            state_rep = torch.repeat_interleave(next_state, n, 0)
            sampled_action = generator.decode(state_rep)
            perturbed_action = tpert(state_rep, sampled_action)
            target_Q1 = tvalue(state_rep, perturbed_action)
            target_Q2 = tvalue(state_rep, perturbed_action) if tvalue.training else target_Q1
        target_value = 0.75 * torch.min(target_Q1, target_Q2)
        target_value += 0.25 * torch.max(target_Q1, target_Q2)
        target_value = target_value.view(batch_size, -1).max(1)[0].view(-1, 1)
        expected_value = temporal_difference(reward, done, params["gamma"], target_value)
    value = nets["value_net1"](state, action)
    value_loss = torch.pow(value - expected_value, 2).mean()
    if learn:
        optimizer["value_optimizer1"].zero_grad()
        optimizer["value_optimizer2"].zero_grad()
        value_loss.backward()
        optimizer["value_optimizer1"].step()
        optimizer["value_optimizer2"].step()
    else:
        if log:
            writer.add_histogram("value", value, step)
            writer.add_histogram("target_value", target_value, step)
            writer.add_histogram("expected_value", expected_value, step)
        writer.close()

    # ---- perturbator -------------------------------------------------------------------------------------------------
    with torch.no_grad():
        sampled_actions = generator.decode(state)
    perturbed_actions = nets["perturbator_net"](state, sampled_actions)
    perturbator_loss = -_score(nets["value_net1"], state, perturbed_actions)
    if not learn and log:
        writer.add_histogram("perturbator_loss", perturbator_loss, step)
    perturbator_loss = perturbator_loss.mean()
    if learn:
        if step % params["perturbator_step"] == 0:
            optimizer["perturbator_optimizer"].zero_grad()
            perturbator_loss.backward()
            torch.nn.utils.clip_grad_norm_(nets["perturbator_net"].parameters(), -1, 1)
            optimizer["perturbator_optimizer"].step()
        utils.soft_update(nets["value_net1"], nets["target_value_net1"], soft_tau=params["soft_tau"])
        utils.soft_update(nets["value_net2"], nets["target_value_net2"], soft_tau=params["soft_tau"])
        utils.soft_update(nets["perturbator_net"], nets["target_perturbator_net"], soft_tau=params["soft_tau"])
    else:
        debug["sampled_actions"] = sampled_actions
        debug["perturbed_actions"] = perturbed_actions
        if log:
            writer.add_figure("sampled_actions", utils.pairwise_distances_fig(sampled_actions[:50]), step)
            writer.add_figure("perturbed_actions", utils.pairwise_distances_fig(perturbed_actions[:50]), step)

    # (inside a stream capture -- recnn_amd.nn.graphed.GraphedUpdate -- the losses stay device scalars: .item() would sync)
    val = (lambda t: t.detach()) if (dev.type == "cuda" and torch.cuda.is_current_stream_capturing()) else (lambda t: t.item())
    losses = {"value": val(value_loss), "perturbator": val(perturbator_loss), "generator": val(generator_loss), "step": step}
    utils.write_losses(writer, losses, kind="train" if learn else "test")
    return losses
