"""REINFORCE updates (reference: recnn/nn/update/reinforce.py:10-129) -- SURVEY.md 8 row f1.

`ChooseREINFORCE` turns the episode a `DiscreteActor` has collected (`rewards`, `saved_log_probs`, `correction`,
`lambda_k`) into the policy loss and steps the policy optimizer; `reinforce_update` is one environment step: act, score the
action distribution with the critic, critic TD update, and every `policy_step` steps the policy update + soft target updates.

The catalogue-sized work (the [B, hidden] x [hidden, n_items] GEMMs, softmax / sampling / log-prob over n_items, their
backward, the optimizer passes over the n_items x hidden weights, the soft updates) runs in the HIP kernels; what is left
to torch here is algebra on per-row vectors of length B and on the `policy_step` episode scalars.

`nets["policy_net"]` / `nets["target_policy_net"]` may be `recnn_amd.parallel.VocabParallelDiscreteActor` and the critics
`VocabParallelCritic` (one process per GPU, the catalogue dimension of the actor's head and of the critic's first layer sharded over
the ranks): the same function body runs on every rank, "probabilities" are then each rank's columns (tests/test_vocab_parallel_gloo.py,
tests/test_gpu_reinforce.py).
"""
import torch

from ... import data, utils
from .misc import value_update

__all__ = ["ChooseREINFORCE", "reinforce_update"]


class ChooseREINFORCE:
    def __init__(self, method=None):
        self.method = ChooseREINFORCE.basic_reinforce if method is None else method

    # The three estimators differ in the per-row weight of -log pi(a|s) * R_t:
    @staticmethod
    def basic_reinforce(policy, returns, *args, **kwargs):
        terms = [-lp * R for lp, R in zip(policy.saved_log_probs, returns)]
        return torch.cat(terms).sum()

    @staticmethod
    def reinforce_with_correction(policy, returns, *args, **kwargs):
        # weight = pi(a|s) / beta(a|s)
        terms = [c * -lp * R for c, lp, R in zip(policy.correction, policy.saved_log_probs, returns)]
        return torch.cat(terms).sum()

    @staticmethod
    def reinforce_with_TopK_correction(policy, returns, *args, **kwargs):
        # weight = lambda_K * pi(a|s) / beta(a|s)
        terms = [lk * c * -lp * R
                 for lk, c, lp, R in zip(policy.lambda_k, policy.correction, policy.saved_log_probs, returns)]
        return torch.cat(terms).sum()

    @staticmethod
    def discounted_returns(rewards, gamma=0.99, eps=0.0001):
        """R_t = r_t + gamma R_{t+1}, then (R - mean) / (std + eps) with the unbiased std (reinforce.py:45-53)."""
        if len(rewards) == 0:
            return torch.zeros(0)
        dev = rewards[0].device if isinstance(rewards[0], torch.Tensor) else "cpu"
        r = torch.stack([torch.as_tensor(x, dtype=torch.float32, device=dev).reshape(()) for x in rewards])
        out = torch.empty_like(r)
        run = torch.zeros((), dtype=torch.float32, device=dev)
        for t in range(r.numel() - 1, -1, -1):
            run = r[t] + gamma * run
            out[t] = run
        return (out - out.mean()) / (out.std() + eps)

    def __call__(self, policy, optimizer, learn=True):
        returns = self.discounted_returns(policy.rewards)
        policy_loss = self.method(policy, returns)
        if learn:
            optimizer.zero_grad()
            if policy_loss.is_cuda:
                from .. import functional as F_hip
                with F_hip.episode_backward():     # one weight-gradient GEMM for the whole episode instead of one (+ an add_) per step
                    policy_loss.backward()
            else:
                policy_loss.backward()
            optimizer.step()
        policy.gc()
        return policy_loss


def reinforce_update(batch, params, nets, optimizer, device=torch.device("cpu"), debug=None, writer=utils.DummyWriter(),
                     learn=True, step=-1):
    learn = True   # REINFORCE has no evaluation mode: every call acts and records (reinforce.py:80-81)
    policy = nets["policy_net"]
    # (the policy's device, not get_base_batch's default "cuda" = cuda:0: reinforce.py:83 passes device=device)
    state, action, reward, next_state, done = data.get_base_batch(batch, device=next(policy.parameters()).device)
    predicted_probs = policy.select_action(state=state, action=action, K=params["K"], learn=learn, writer=writer, step=step)
    if not isinstance(writer, utils.DummyWriter):
        mx = predicted_probs.max(dim=1).values
        writer.add_histogram("predicted_probs_std", predicted_probs.std(), step)
        writer.add_histogram("predicted_probs_mean", predicted_probs.mean(), step)
        writer.add_histogram("predicted_probs_max_mean", mx.mean(), step)
        writer.add_histogram("predicted_probs_max_std", mx.std(), step)
    # the critic scores the whole action distribution; its mean over the batch is this step's reward
    with torch.no_grad():
        policy.rewards.append(nets["value_net"](state, predicted_probs).mean())
    value_loss = value_update(batch, params, nets, optimizer, writer=writer, device=device, debug=debug, learn=True, step=step)
    if step % params["policy_step"] == 0 and step > 0:
        policy_loss = params["reinforce"](policy, optimizer["policy_optimizer"])
        utils.soft_update(nets["value_net"], nets["target_value_net"], soft_tau=params["soft_tau"])
        utils.soft_update(policy, nets["target_policy_net"], soft_tau=params["soft_tau"])
        losses = {"value": value_loss.item(), "policy": policy_loss.item(), "step": step}
        utils.write_losses(writer, losses, kind="train" if learn else "test")
        return losses
