"""Actor and critic networks of the MADDPG path, same parameter names and shapes as the reference so
state_dicts interchange (actor 34 948 params, critic 52 097 params at the default config).
They run through PyTorch-ROCm (rocBLAS GEMMs); at batch = thousands of envs x 5 agents x 64-wide layers
these are plain library GEMMs, not a custom-kernel target."""
from __future__ import annotations

import torch as th
import torch.nn as nn
import torch.nn.functional as F


def _activation(name):
    if name == "relu":
        return F.relu
    if name == "tanh":
        return th.tanh
    raise ValueError(name)


class RNNAgent(nn.Module):
    """madrl/agents/rnn_agent.py:13-33: fc1 -> LayerNorm -> act -> GRUCell -> fc2; forward returns
    (action mean, None, new hidden)."""

    def __init__(self, input_shape, args):
        super().__init__()
        self.args = args
        self.fc1 = nn.Linear(input_shape, args.hid_size)
        if args.layernorm:
            self.layernorm = nn.LayerNorm(args.hid_size)
        self.rnn = nn.GRUCell(args.hid_size, args.hid_size)
        self.fc2 = nn.Linear(args.hid_size, args.action_dim)
        self._act = _activation(args.hid_activation)

    def init_hidden(self):
        return self.fc1.weight.new_zeros(1, self.args.agent_num, self.args.hid_size)

    def forward(self, inputs, hidden_state):
        x = self.fc1(inputs)
        if self.args.layernorm:
            x = self.layernorm(x)
        h = self.rnn(self._act(x), hidden_state.reshape(-1, self.args.hid_size))
        return self.fc2(h), None, h


class MLPAgent(nn.Module):
    """madrl/agents/mlp_agent.py:5-32 (agent_type: mlp)."""

    def __init__(self, input_shape, args):
        super().__init__()
        self.args = args
        self.fc1 = nn.Linear(input_shape, args.hid_size)
        if args.layernorm:
            self.layernorm = nn.LayerNorm(args.hid_size)
        self.fc2 = nn.Linear(args.hid_size, args.hid_size)
        self.fc3 = nn.Linear(args.hid_size, args.action_dim)
        self._act = _activation(args.hid_activation)

    def init_hidden(self):
        return self.fc1.weight.new_zeros(1, self.args.hid_size)

    def forward(self, inputs, hidden_state):
        x = self.fc1(inputs)
        if self.args.layernorm:
            x = self.layernorm(x)
        h = self._act(self.fc2(self._act(x)))
        return self.fc3(h), None, h


class MLPCritic(nn.Module):
    """madrl/critics/mlp_critic.py:5-34: fc1 -> LayerNorm -> act -> fc2 -> act -> fc3; returns (v, h).

    ``forward_from_hidden`` takes the pre-LayerNorm first-layer activation directly: the centralised
    critic's input repeats every agent's observation n times (maddpg.py:38-39), so the caller forms
    fc1's output from its column blocks once per sample instead of n times (learner.MADDPG.value)."""

    def __init__(self, input_shape, output_shape, args):
        super().__init__()
        self.args = args
        self.fc1 = nn.Linear(input_shape, args.hid_size)
        if args.layernorm:
            self.layernorm = nn.LayerNorm(args.hid_size)
        self.fc2 = nn.Linear(args.hid_size, args.hid_size)
        self.fc3 = nn.Linear(args.hid_size, output_shape)
        self._act = _activation(args.hid_activation)

    def init_hidden(self):
        return self.fc1.weight.new_zeros(1, self.args.hid_size)

    def forward_from_hidden(self, x):
        if self.args.layernorm:
            x = self.layernorm(x)
        h = self._act(self.fc2(self._act(x)))
        return self.fc3(h), h

    def forward(self, inputs, hidden_state):
        return self.forward_from_hidden(self.fc1(inputs))
