"""Actor / Critic / DiscreteActor / bcqPerturbator / bcqGenerator
(reference: recnn/nn/models.py:41-73, :187-213, :76-184, :216-242, :245-295) and the Soft Actor-Critic networks of the
reference's `examples/1. Vanilla RL/4. SAC.ipynb` (StateCritic, SoftQ, StochasticActor: code cells 5-7).

Real nn.Modules with the reference's sub-module names (linear1/2/3, drop_layer), state_dict keys and
constructor RNG consumption (nn.Linear default init for linear1, linear2, linear3 in that order, then
linear3.weight/bias ~ U(-init_w, init_w)), so equal seeds give equal initial weights.  Called directly they
run the HIP GEMM kernels through `recnn_amd.nn.functional`; inside `ddpg_update` / `td3_update` their
parameters are adopted by the fused step engine (the parameter tensors become views into its flat arenas).
"""
import math

import torch
import torch.nn as nn

from . import functional as F_hip

__all__ = ["Actor", "Critic", "DiscreteActor", "Beta", "bcqPerturbator", "bcqGenerator", "StateCritic", "SoftQ", "StochasticActor"]


class Actor(nn.Module):
    """state -> action: relu(L1) -> dropout(0.5) -> relu(L2) -> dropout(0.5) -> L3 [-> tanh]."""

    def __init__(self, input_dim, action_dim, hidden_size, init_w=2e-1):
        super().__init__()
        self.drop_layer = nn.Dropout(p=0.5)
        self.linear1 = nn.Linear(input_dim, hidden_size)
        self.linear2 = nn.Linear(hidden_size, hidden_size)
        self.linear3 = nn.Linear(hidden_size, action_dim)
        self.linear3.weight.data.uniform_(-init_w, init_w)
        self.linear3.bias.data.uniform_(-init_w, init_w)

    def forward(self, state, tanh=False):
        action = F_hip.mlp(state, self, self.training)
        return torch.tanh(action) if tanh else action


class Critic(nn.Module):
    """(state, action) -> value: the same MLP over cat([state, action], 1) with a single output."""

    def __init__(self, input_dim, action_dim, hidden_size, init_w=3e-5):
        super().__init__()
        self.drop_layer = nn.Dropout(p=0.5)
        self.linear1 = nn.Linear(input_dim + action_dim, hidden_size)
        self.linear2 = nn.Linear(hidden_size, hidden_size)
        self.linear3 = nn.Linear(hidden_size, 1)
        self.linear3.weight.data.uniform_(-init_w, init_w)
        self.linear3.bias.data.uniform_(-init_w, init_w)

    def forward(self, state, action):
        idx = F_hip.onehot_index_of(action)
        if idx is not None and state.is_cuda:
            # one-hot action rows (REINFORCE's discrete-action batches, data.batch_contstate_discaction): gather the B
            # weight columns instead of contracting over the whole catalogue
            return F_hip.mlp_onehot(state, idx, state.shape[1], self, self.training)
        return F_hip.mlp(torch.cat([state, action], 1), self, self.training)

    def candidates(self, state, actions, n):
        """self(repeat_interleave(state, n, 0), actions) for n consecutive candidate actions per state row, without
        dropout or gradient (`functional.mlp_candidates`: the state part of layer 1 once per state)."""
        return F_hip.mlp_candidates(state, actions, n, self.linear1.weight, self.linear1.bias, self.linear2.weight,
                                    self.linear2.bias, self.linear3.weight, self.linear3.bias)


class Beta(nn.Module):
    """The learned behaviour policy of the Top-K off-policy correction -- the `Beta` class the reference DEFINES IN ITS NOTEBOOK
    (`examples/2. REINFORCE TopK Off Policy Correction/3. TopK Reinforce Off Policy Correction.ipynb`, cell 3) and hands to
    `DiscreteActor._select_action_with_TopK_correction` as `beta_net.forward` (recnn/nn/models.py:113-141,143-184):

        self.net = Sequential(Linear(1290, num_items), Softmax());  optim = RAdam(lr=1e-5, weight_decay=1e-5);  CrossEntropyLoss
        forward(state, action):  p = net(state);  loss = criterion(p, action.argmax(1));  zero_grad / backward / optim.step();
                                 return p.detach()

    i.e. it TRAINS on every call (one optimizer step on the cross entropy of its own probabilities taken as logits -- the
    notebook's quirk, kept) and returns the probabilities it had BEFORE that step.  Same sub-module names (`net.0.weight`,
    `net.0.bias` in the state dict), same attributes `optim` / `criterion`.  Forward, loss, backward and optimizer run on the
    HIP kernels (`functional.beta_train_forward`, `recnn_amd.optim`).  Optimizer: the notebook's `torch_optimizer.RAdam` is an
    absent, un-pinned package; the default here follows `recnn_amd.nn.algo.set_default_optimizer` (this package's fused Ranger
    or Adam, lr = 1e-5, weight_decay = 1e-5), `optimizer=lambda params: ...` injects any other (the parity fixture injects
    torch.optim.Adam).  `last_loss` keeps the loss of the latest call as a device scalar."""

    def __init__(self, input_dim=1290, num_items=5000, lr=1e-5, weight_decay=1e-5, optimizer=None):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(input_dim, num_items), nn.Softmax(dim=1))
        self.criterion = nn.CrossEntropyLoss()
        self._opt_args = (lr, weight_decay, optimizer)
        self.optim = None
        self.last_loss = None

    def _optimizer(self):
        if self.optim is None:
            from .. import optim as O
            from . import algo
            lr, wd, make = self._opt_args
            if make is not None:
                self.optim = make(self.net.parameters())
            else:
                cls = O.Ranger if algo._DEFAULT_OPTIMIZER == "ranger" else O.Adam
                self.optim = cls(self.net.parameters(), lr=lr, weight_decay=wd)
        return self.optim

    def forward(self, state, action=None):
        lin = self.net[0]
        if action is None:                      # plain evaluation (no label, nothing to learn from)
            with torch.no_grad():
                tgt = torch.zeros(state.shape[0], dtype=torch.int64, device=state.device)
                p, _, _, _ = F_hip.beta_train_forward(state, tgt, lin.weight, lin.bias)
            return p
        idx = F_hip.onehot_index_of(action)
        tgt = idx if idx is not None else action.argmax(1)
        p, loss, gw, gb = F_hip.beta_train_forward(state, tgt, lin.weight, lin.bias)
        opt = self._optimizer()
        lin.weight.grad, lin.bias.grad = gw, gb          # (zero_grad + backward of the notebook)
        opt.step()
        self.last_loss = loss
        return p


class DiscreteActor(nn.Module):
    """state -> probabilities over the catalogue: softmax(L2(relu(L1(state))))  (models.py:76-184), the REINFORCE policy.

    Same attributes as the reference: `saved_log_probs`, `rewards`, `correction`, `lambda_k` (the episode the update
    functions append to and `ChooseREINFORCE` consumes), `action_source` ({"pi": "pi", "beta": "beta"}; set pi to "beta"
    to score the behaviour policy's action, issue #7 of the reference) and the re-bindable `select_action`.
    Forward, sampling and log-prob are one autograd node on the HIP kernels (`functional.DiscretePolicyFunction`); the
    sampler is an inverse-CDF walk with a counter-based uniform per row (torch's multinomial stream is not reproduced --
    equal in distribution, tests/test_gpu_reinforce.py).  `forced_actions` (a list of int64 tensors, consumed first-in
    first-out) replaces the next draws of the policy: replaying logged actions, and the parity tests.
    """

    def __init__(self, input_dim, action_dim, hidden_size, init_w=0):
        super().__init__()
        self.linear1 = nn.Linear(input_dim, hidden_size)
        self.linear2 = nn.Linear(hidden_size, action_dim)
        self.saved_log_probs = []
        self.rewards = []
        self.correction = []
        self.lambda_k = []
        self.action_source = {"pi": "pi", "beta": "beta"}
        self.select_action = self._select_action
        self.forced_actions = []

    def forward(self, inputs):
        probs, _, _ = F_hip.discrete_policy(inputs, self)
        return probs

    def gc(self):
        del self.rewards[:]
        del self.saved_log_probs[:]
        del self.correction[:]
        del self.lambda_k[:]

    def _act(self, state, actions=None):
        """(probs, action, log_prob); draws the action unless one is given or queued in `forced_actions`."""
        if actions is None and self.forced_actions:
            actions = self.forced_actions.pop(0)
        return F_hip.discrete_policy(state, self, actions=actions, sample=actions is None)

    def _select_action(self, state, **kwargs):
        # plain REINFORCE: there is no behaviour policy, `action_source` is not consulted (models.py:103-111)
        pi_probs, _, log_prob = self._act(state)
        self.saved_log_probs.append(log_prob)
        return pi_probs

    def pi_beta_sample(self, state, beta, action, **kwargs):
        """log-probs of the target (pi) and behaviour (beta) policies for one action each (models.py:113-141).
        `beta(state, action=...)` returns the behaviour policy's probabilities.  The HIP sampler (`F_hip.categorical`) is
        not differentiable: for a beta that detaches its output (the reference notebook's `Beta` does) nothing is lost; a
        beta that returns probabilities still on the autograd graph gets its log-prob recomputed with torch ops below so
        that the gradient through `corr = exp(pi_lp) / exp(beta_lp)` reaches it, as in the reference."""
        beta_probs = beta(state.detach(), action=action)
        beta_action, beta_log_prob_own = F_hip.categorical(beta_probs)
        if beta_probs.requires_grad:
            def _lp(act):   # Categorical(probs).log_prob(act): probs normalised, clamped to [eps, 1 - eps], log, gather
                pr = beta_probs / beta_probs.sum(-1, keepdim=True)
                eps = torch.finfo(pr.dtype).eps
                return torch.log(pr.clamp(eps, 1 - eps)).gather(-1, act.reshape(-1, 1).to(torch.int64)).reshape(act.shape)
            beta_log_prob_own = _lp(beta_action)
        if self.action_source["pi"] == "beta":
            pi_probs, pi_action, pi_log_prob = self._act(state, actions=beta_action)
        else:
            pi_probs, pi_action, pi_log_prob = self._act(state)
        if self.action_source["beta"] == "beta":
            beta_log_prob = beta_log_prob_own
        elif beta_probs.requires_grad:
            beta_log_prob = _lp(pi_action)
        else:
            _, beta_log_prob = F_hip.categorical(beta_probs, actions=pi_action)
        return pi_log_prob, beta_log_prob, pi_probs

    def _select_action_with_correction(self, state, beta, action, writer, step, **kwargs):
        pi_log_prob, beta_log_prob, pi_probs = self.pi_beta_sample(state, beta, action)
        corr = torch.exp(pi_log_prob) / torch.exp(beta_log_prob)      # off-policy importance weight
        writer.add_histogram("correction", corr, step)
        writer.add_histogram("pi_log_prob", pi_log_prob, step)
        writer.add_histogram("beta_log_prob", beta_log_prob, step)
        self.correction.append(corr)
        self.saved_log_probs.append(pi_log_prob)
        return pi_probs

    def _select_action_with_TopK_correction(self, state, beta, action, K, writer, step, **kwargs):
        pi_log_prob, beta_log_prob, pi_probs = self.pi_beta_sample(state, beta, action)
        corr = torch.exp(pi_log_prob) / torch.exp(beta_log_prob)
        l_k = K * (1 - torch.exp(pi_log_prob)) ** (K - 1)             # lambda_K = K (1 - pi)^(K-1)  (models.py:173)
        writer.add_histogram("correction", corr, step)
        writer.add_histogram("l_k", l_k, step)
        writer.add_histogram("pi_log_prob", pi_log_prob, step)
        writer.add_histogram("beta_log_prob", beta_log_prob, step)
        self.correction.append(corr)
        self.lambda_k.append(l_k)
        self.saved_log_probs.append(pi_log_prob)
        return pi_probs


class bcqPerturbator(nn.Module):
    """(state, action) -> action + MLP([state | action]): BCQ's perturbation network (models.py:216-242).  Same sub-module
    names, state_dict keys and constructor RNG consumption as the reference; forward / backward on the HIP GEMM kernels."""

    def __init__(self, num_inputs, num_actions, hidden_size, init_w=3e-1):
        super().__init__()
        self.drop_layer = nn.Dropout(p=0.5)
        self.linear1 = nn.Linear(num_inputs + num_actions, hidden_size)
        self.linear2 = nn.Linear(hidden_size, hidden_size)
        self.linear3 = nn.Linear(hidden_size, num_actions)
        self.linear3.weight.data.uniform_(-init_w, init_w)
        self.linear3.bias.data.uniform_(-init_w, init_w)

    def forward(self, state, action):
        return F_hip.mlp(torch.cat([state, action], 1), self, self.training) + action

    def candidates(self, state, actions, n):
        """self(repeat_interleave(state, n, 0), actions) without dropout or gradient (see Critic.candidates)."""
        return F_hip.mlp_candidates(state, actions, n, self.linear1.weight, self.linear1.bias, self.linear2.weight,
                                    self.linear2.bias, self.linear3.weight, self.linear3.bias) + actions


class bcqGenerator(nn.Module):
    """BCQ's conditional VAE (models.py:245-295): encoder e1, e2 -> (mean, log_std), z = mean + std * eps, decoder d1, d2, d3
    over [state | z].  `forward` returns (reconstructed action, mean, std); `decode(state)` draws z ~ clamp(N(0, 1), +-0.5).

    The encoder and the decoder each run as one 3-layer stack on the HIP GEMM kernels (the `mean` and `log_std` heads share
    one [2 latent, 750] GEMM); clamp / exp / reparametrisation are one kernel (csrc/vae.hip).  The reference draws its
    normals on the CPU from the global generator and copies them over; here they are drawn on the device
    (`torch.randn(device=...)`) -- equal in distribution.  `forced_noise` (a list of tensors, consumed first-in first-out)
    replaces the next draws: replaying logged noise, and the parity tests."""

    def __init__(self, state_dim, action_dim, latent_dim):
        super().__init__()
        self.e1 = nn.Linear(state_dim + action_dim, 750)
        self.e2 = nn.Linear(750, 750)
        self.mean = nn.Linear(750, latent_dim)
        self.log_std = nn.Linear(750, latent_dim)
        self.d1 = nn.Linear(state_dim + latent_dim, 750)
        self.d2 = nn.Linear(750, 750)
        self.d3 = nn.Linear(750, action_dim)
        self.latent_dim = latent_dim
        self.normal = torch.distributions.Normal(0, 1)
        self.forced_noise = []

    def _noise(self, rows, device):
        if self.forced_noise:
            z = self.forced_noise.pop(0)
            if tuple(z.shape) != (rows, self.latent_dim):
                raise ValueError(f"forced_noise entry has shape {tuple(z.shape)}, the call needs {(rows, self.latent_dim)}")
            return z.to(device=device, dtype=torch.float32)
        return torch.randn(rows, self.latent_dim, device=device)

    def forward(self, state, action):
        ml = F_hip.mlp3(torch.cat([state, action], 1), self.e1, self.e2,
                        torch.cat([self.mean.weight, self.log_std.weight], 0), torch.cat([self.mean.bias, self.log_std.bias], 0))
        z, std = F_hip.vae_latent(ml, self._noise(state.shape[0], state.device))
        return self.decode(state, z), ml[:, :self.latent_dim], std

    def decode(self, state, z=None):
        if z is None:
            z = self._noise(state.shape[0], state.device).clamp(-0.5, 0.5)
        return F_hip.mlp3(torch.cat([state, z], 1), self.d1, self.d2, self.d3.weight, self.d3.bias)

    def decode_candidates(self, state, n):
        """decode(repeat_interleave(state, n, 0)): n sampled actions per state row, [B n, action_dim], no gradient."""
        z = self._noise(state.shape[0] * n, state.device).clamp(-0.5, 0.5)
        return F_hip.mlp_candidates(state, z, n, self.d1.weight, self.d1.bias, self.d2.weight, self.d2.bias,
                                    self.d3.weight, self.d3.bias)


# ---- Soft Actor-Critic (examples/1. Vanilla RL/4. SAC.ipynb, code cells 5-7) ------------------------
class StateCritic(nn.Module):
    """state -> V(state): relu(L1) -> relu(L2) -> L3, no dropout (SAC.ipynb cell 5)."""

    def __init__(self, state_dim, hidden_dim, init_w=3e-3):
        super().__init__()
        self.linear1 = nn.Linear(state_dim, hidden_dim)
        self.linear2 = nn.Linear(hidden_dim, hidden_dim)
        self.linear3 = nn.Linear(hidden_dim, 1)
        self.linear3.weight.data.uniform_(-init_w, init_w)
        self.linear3.bias.data.uniform_(-init_w, init_w)

    def forward(self, state):
        return F_hip.mlp(state, self, False)


class SoftQ(nn.Module):
    """(state, action) -> Q: the same stack over cat([state, action], 1), no dropout (SAC.ipynb cell 6)."""

    def __init__(self, input_dim, action_dim, hidden_dim, init_w=3e-3):
        super().__init__()
        self.linear1 = nn.Linear(input_dim + action_dim, hidden_dim)
        self.linear2 = nn.Linear(hidden_dim, hidden_dim)
        self.linear3 = nn.Linear(hidden_dim, 1)
        self.linear3.weight.data.uniform_(-init_w, init_w)
        self.linear3.bias.data.uniform_(-init_w, init_w)

    def forward(self, state, action):
        return F_hip.mlp(torch.cat([state, action], 1), self, False)


class StochasticActor(nn.Module):
    """state -> (mean, log_std) of a tanh-squashed Gaussian policy (SAC.ipynb cell 7): relu(L1) -> dropout -> relu(L2) ->
    dropout -> {mean_linear, clamp(log_std_linear)}.  The trunk and BOTH heads run as one three-layer HIP MLP (the heads'
    weights concatenated to a [2 * action_dim, hidden] last layer; autograd splits the gradient back).

    Kept from the notebook: `evaluate` draws ONE standard-normal scalar `z` per call (`Normal(0, 1).sample()` has no batch
    shape) shared by every row and action dimension, evaluates `Normal(mean, std).log_prob` at the SQUASHED action, and does
    not sum the log-prob over the action dimension ([B, action_dim]).  `forced_z` (a list, consumed first-in first-out)
    replaces the next draws: replays and parity tests."""

    def __init__(self, input_dim, action_dim, hidden_size, params):
        super().__init__()
        self.log_std_min = params["log_std_min"]
        self.log_std_max = params["log_std_max"]
        self.drop_layer = nn.Dropout(p=0.5)
        self.linear1 = nn.Linear(input_dim, hidden_size)
        self.linear2 = nn.Linear(hidden_size, hidden_size)
        self.mean_linear = nn.Linear(hidden_size, action_dim)
        self.mean_linear.weight.data.uniform_(-params["mean_initw"], params["mean_initw"])
        self.mean_linear.bias.data.uniform_(-params["mean_initw"], params["mean_initw"])
        self.log_std_linear = nn.Linear(hidden_size, action_dim)
        self.log_std_linear.weight.data.uniform_(-params["std_initw"], params["std_initw"])
        self.log_std_linear.bias.data.uniform_(-params["std_initw"], params["std_initw"])
        self.forced_z = []

    def forward(self, state):
        w3 = torch.cat([self.mean_linear.weight, self.log_std_linear.weight], 0)
        b3 = torch.cat([self.mean_linear.bias, self.log_std_linear.bias], 0)
        out = F_hip.MLPFunction.apply(state.float(), self.linear1.weight, self.linear1.bias, self.linear2.weight, self.linear2.bias,
                                      w3, b3, self.training, torch.initial_seed(), F_hip._take_forced_masks(self, self.training), None)
        a = self.mean_linear.out_features
        return out[:, :a], torch.clamp(out[:, a:], self.log_std_min, self.log_std_max)

    def evaluate(self, state, epsilon=1e-6):
        mean, log_std = self.forward(state)
        std = log_std.exp()
        # (drawn on the device: no host value enters the step, so it can be captured -- recnn_amd.nn.GraphedUpdate)
        z = self.forced_z.pop(0) if self.forced_z else torch.randn((), device=mean.device)
        z = torch.as_tensor(z, dtype=torch.float32).to(mean.device)
        action = torch.tanh(mean + z * std)
        # torch.distributions.Normal(mean, std).log_prob(action), its formula written out (the class validates its arguments with
        # a host sync, which a captured step cannot have)
        log_prob = -((action - mean) ** 2) / (2 * std ** 2) - std.log() - math.log(math.sqrt(2 * math.pi))
        log_prob = log_prob - torch.log(1 - action.pow(2) + epsilon)
        return action, log_prob, z, mean, log_std
