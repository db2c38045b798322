"""Actor / Critic (reference: recnn/nn/models.py:41-73, :187-213).

Real nn.Modules with the reference's sub-module names (linear1/2/3, drop_layer), state_dict keys and
constructor RNG consumption (nn.Linear default init for linear1, linear2, linear3 in that order, then
linear3.weight/bias ~ U(-init_w, init_w)), so equal seeds give equal initial weights.  Called directly they
run the HIP GEMM kernels through `recnn_amd.nn.functional`; inside `ddpg_update` / `td3_update` their
parameters are adopted by the fused step engine (the parameter tensors become views into its flat arenas).
"""
import torch
import torch.nn as nn

from . import functional as F_hip

__all__ = ["Actor", "Critic"]


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
        return F_hip.mlp(torch.cat([state, action], 1), self, self.training)
