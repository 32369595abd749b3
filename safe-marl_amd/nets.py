"""Actor and critic networks of the MADDPG path, same parameter names and shapes as the reference so
state_dicts interchange (actor 34 948 params, critic 52 097 params at the default config).
Training (autograd) runs through PyTorch-ROCm; the actor's INFERENCE pass — every rollout step and every
bootstrap target — is one fused HIP launch (csrc/actor.hip, include/flexnet.h) through ``fused_actor_forward``."""
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


def fused_actor_forward(agent, obs, hidden, n_agents, agent_id):
    """rnn_agent.py:25-33 + model.py:102-116 without an autograd graph, in one HIP launch.

    ``obs`` [b, n, obs_dim] fp32 on the GPU WITHOUT the one-hot id columns (the kernel adds fc1's id column of
    row r % n itself), ``hidden`` [b, n, 64] or [b * n, 64].  Returns (means [b * n, act], hidden [b * n, 64]) or
    None when the configuration is not one the kernel covers (the caller then uses the module)."""
    import ctypes as C
    from . import _lib
    a = agent.args
    if not (isinstance(agent, RNNAgent) and a.hid_size == 64 and a.hid_activation == "relu" and obs.is_cuda
            and obs.dtype == th.float32 and obs.shape[-1] <= 144 and n_agents <= 8 and a.action_dim <= 8):
        return None
    lib = _lib.load()
    rows = obs.shape[0] * obs.shape[1]
    obs = obs.contiguous()
    hidden = hidden.reshape(rows, 64).to(th.float32).contiguous()
    means = th.empty(rows, a.action_dim, dtype=th.float32, device=obs.device)
    hid_out = th.empty(rows, 64, dtype=th.float32, device=obs.device)
    args = _lib.FlexActorArgs()
    args.rows, args.n_agents, args.obs_dim, args.act_dim = rows, n_agents, obs.shape[-1], a.action_dim
    args.agent_id, args.layernorm, args.ln_eps = int(bool(agent_id)), int(bool(a.layernorm)), 1e-5
    ln = agent.layernorm if a.layernorm else None
    if ln is not None:
        args.ln_eps = float(ln.eps)
    for name, t in (("obs", obs), ("hidden_in", hidden), ("fc1_w", agent.fc1.weight), ("fc1_b", agent.fc1.bias),
                    ("ln_w", ln.weight if ln is not None else None), ("ln_b", ln.bias if ln is not None else None),
                    ("w_ih", agent.rnn.weight_ih), ("w_hh", agent.rnn.weight_hh), ("b_ih", agent.rnn.bias_ih),
                    ("b_hh", agent.rnn.bias_hh), ("fc2_w", agent.fc2.weight), ("fc2_b", agent.fc2.bias),
                    ("means", means), ("hidden_out", hid_out)):
        if t is not None and not t.is_contiguous():
            return None
        setattr(args, name, None if t is None else t.data_ptr())
    rc = lib.flexnet_actor_forward(C.byref(args), C.c_void_p(th.cuda.current_stream().cuda_stream))
    if rc == _lib.FLEXNET_EUNSUPPORTED:
        return None
    _lib.check(rc, "flexnet_actor_forward")
    return means, hid_out
