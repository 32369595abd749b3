"""Actor and critic networks of the MADDPG path, same parameter names and shapes as the reference so
state_dicts interchange (actor 34 948 params, critic 52 097 params at the default config).
Training (autograd) runs through PyTorch-ROCm; the actor's INFERENCE pass — every rollout step and every
bootstrap target — is one fused HIP launch (csrc/actor.hip, include/flexnet.h) through ``fused_actor_forward``."""
from __future__ import annotations

import os
import torch as th
import torch.nn as nn
import torch.nn.functional as F

from .util import note_fallback


def _find_fused_gru():
    """ATen's pointwise GRU cell, what nn.GRUCell dispatches to on the GPU after its two gate GEMMs (ATen RNN.cpp)."""
    try:
        return th.ops.aten._thnn_fused_gru_cell.default
    except (AttributeError, RuntimeError):
        return None


_FUSED_GRU = _find_fused_gru()


def _gru_cell(x, hx, rnn):
    """nn.GRUCell as ATen composes it on the GPU, for update batches: the biases ride the gate GEMMs (their gradients
    are then column sums taken inside csrc/wgrad.hip's pass over the gate gradients) instead of entering the pointwise
    cell, whose backward would reduce them with two more passes over [rows, 192]."""
    gi = tall_linear(x, rnn.weight_ih, rnn.bias_ih)
    gh = tall_linear(hx, rnn.weight_hh, rnn.bias_hh)
    return _FUSED_GRU(gi, gh, hx, None, None)[0]


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
        if th.is_grad_enabled() and inputs.is_cuda and not inputs.requires_grad and inputs.dim() == 2:
            x = wide_batch_linear(inputs, self.fc1.weight) + self.fc1.bias     # split-K weight gradient at update batches
        else:
            x = self.fc1(inputs)
        if self.args.layernorm:
            x = self.layernorm(x)
        x, hx = self._act(x), hidden_state.reshape(-1, self.args.hid_size)
        if (th.is_grad_enabled() and x.is_cuda and x.dim() == 2 and x.shape[0] >= WGRAD_MIN_ROWS
                and x.dtype == th.float32 and _FUSED_GRU is not None):
            # update batches (163 840 rows): the GRUCell as PyTorch composes it on the GPU — two gate GEMMs and the
            # fused pointwise cell — with the three weight gradients (W_ih, W_hh, fc2) from csrc/wgrad.hip
            r = self.rnn
            h = _gru_cell(x, hx, r)
            return tall_linear(h, self.fc2.weight, self.fc2.bias), None, h
        h = self.rnn(x, hx)
        return self.fc2(h), None, h

    def forward_update(self, obs, hidden_state, n_agents, agent_id):
        """The same forward for an update batch on the GPU, from observations WITHOUT the one-hot id columns: row r is
        agent r % n_agents (model.py:110-112), fc1's id block enters as a per-agent addend.  Returns None when the
        configuration is outside what csrc/lnrelu.hip covers (the caller concatenates the ids and calls forward)."""
        o = obs.shape[1]
        W = self.fc1.weight
        hx0 = hidden_state.reshape(-1, self.args.hid_size)
        if actor_train_supported(self, obs, n_agents, agent_id) and not hx0.requires_grad and hx0.dtype == th.float32:
            # one autograd node: fused matrix-core forward, hand-written backward (csrc/actor.hip, gru.hip, wgrad.hip, lnrelu.hip)
            ln = self.layernorm if self.args.layernorm else None
            r = self.rnn
            means, h = _ActorTrainFn.apply(obs, hx0, n_agents, bool(agent_id), W, self.fc1.bias,
                                           None if ln is None else ln.weight, None if ln is None else ln.bias,
                                           1e-5 if ln is None else ln.eps, r.weight_ih, r.weight_hh, r.bias_ih, r.bias_hh,
                                           self.fc2.weight, self.fc2.bias)
            return means, None, h
        if not (obs.is_cuda and obs.dtype == th.float32 and lnrelu_supported(self, n_agents) and _FUSED_GRU is not None
                and W.shape[1] == o + (n_agents if agent_id else 0) and obs.shape[0] % n_agents == 0):
            if obs.is_cuda:
                note_fallback("actor_update_pass", f"hid {self.args.hid_size}, act {self.args.hid_activation}, agents "
                                                   f"{n_agents}, dtype {obs.dtype}")
            return None
        z = wide_batch_linear(obs, W[:, :o]) if not obs.requires_grad else tall_linear(obs, W[:, :o])
        ln = self.layernorm if self.args.layernorm else None
        x = _LnReluFn.apply(z, self.fc1.bias, W[:, o:].t() if agent_id else None, None if ln is None else ln.weight,
                            None if ln is None else ln.bias, 1e-5 if ln is None else ln.eps, n_agents)
        r, hx = self.rnn, hidden_state.reshape(-1, self.args.hid_size)
        h = _gru_cell(x, hx, r)
        return tall_linear(h, self.fc2.weight, self.fc2.bias), None, h


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

    def forward_from_hidden(self, x, need_hidden=True):
        """``need_hidden=False`` (the MADDPG losses only use v): on the GPU the whole tail — LayerNorm, ReLU, fc2, ReLU,
        fc3 and, when a graph is recorded, their backward — is two HIP launches (csrc/critic.hip)."""
        if not need_hidden and critic_tail_supported(self, x):
            return CriticTail.apply(x, self), None
        if self.args.layernorm:
            x = self.layernorm(x)
        h = self._act(self.fc2(self._act(x)))
        return self.fc3(h), h

    def forward(self, inputs, hidden_state):
        return self.forward_from_hidden(self.fc1(inputs))


# FlexActorArgs.variant when the caller names none (include/flexnet.h: 0 = the library's choice by size; tests pin 2 / 3)
ACTOR_VARIANT = 0


def fused_actor_forward(agent, obs, hidden, n_agents, agent_id, noise=None, std=1.0, low=0.0, high=1.0, variant=None,
                        rng_state=None, ring_cursor=None, obs_slab_stride=0, hid_slab_stride=0, cursor_out=None,
                        out=None, ring_slabs=0, launch=None, obs_source=None):
    """rnn_agent.py:25-33 + model.py:102-116 without an autograd graph, in one HIP launch.

    ``obs`` [b, n, obs_dim] fp32 on the GPU WITHOUT the one-hot id columns (the kernel adds fc1's id column of
    row r % n itself), ``hidden`` [b, n, 64] or [b * n, 64].  Returns (means [b * n, act], hidden [b * n, 64]) or
    None when the configuration is not one the kernel covers (the caller then uses the module).  With ``noise``
    [b, n, act] (standard normal draws) the exploration epilogue runs in the same launch and two more tensors come
    back: action = tanh(mean + std * noise) (util.py:57-64) and the environment's action (util.py:125-128).  With
    ``rng_state`` (int64 device tensor [seed, step]) instead of ``noise`` the kernel draws the normal numbers itself
    (Philox4x32-10 + Box-Muller, csrc/actor.hip actor_noise4); the caller advances ``rng_state[1]`` per call —
    flexnet_rollout_pack does when handed the same tensor.  With ``ring_cursor`` (int64 device tensor) ``obs`` and ``hidden``
    are slab 0 of two slab rings and the launch reads slab ``ring_cursor[0]`` of each (strides in floats), resolved on the
    device — the rollout graph's way of reading the observation where the environment kernel wrote it.
    ``launch`` (with ``ring_slabs``): called with the filled FlexActorArgs INSTEAD of flexnet_actor_forward — the rollout
    burst, which runs this policy evaluation inside the environment's persistent launch (flex_env.rollout_burst).
    ``obs_source`` (a ``_lib.FlexObsSource``, VecFlexProvisionEnv.obs_source()): the observations are read IN PLACE from
    the environment's history (``obs`` then only gives the shape [b, n, obs_dim]; it is not read)."""
    import ctypes as C
    from . import _lib
    a = agent.args
    if not (isinstance(agent, RNNAgent) and a.hid_size == 64 and a.hid_activation == "relu" and obs.is_cuda
            and obs.dtype == th.float32 and obs.shape[-1] <= 144 and n_agents <= 8 and a.action_dim <= 8):
        if obs.is_cuda:
            note_fallback("actor_forward", f"{type(agent).__name__}, hid {a.hid_size}, act {a.hid_activation}, obs "
                                           f"{obs.shape[-1]}, agents {n_agents}, dtype {obs.dtype}")
        return None
    lib = _lib.load()
    rows = obs.shape[0] * obs.shape[1]
    obs_dim = obs.shape[-1]
    obs = obs.contiguous()
    rv = ring_view_of(obs)
    if rv is not None:
        # a placeholder of the replay's stacked-observation ring (RING_VIEWS): the observations are read in place — the ring's
        # base plus, through the kernel's cursor mechanism, the window's first row from the device cell
        if ring_cursor is not None or obs_source is not None or rv[0].shape[1] != obs.shape[1] * obs.shape[2]:
            raise RuntimeError("fused_actor_forward: an in-place window of the stacked-observation ring cannot be combined with a ring cursor")
        ring_cursor, obs_slab_stride, hid_slab_stride, cursor_out = rv[1], rv[0].shape[1], 0, None
        obs = rv[0]
    hidden = hidden.reshape(rows, 64).to(th.float32).contiguous()
    rvh = ring_view_of(hidden)
    if rvh is not None:
        # the hidden states of the same window, in place from the replay's hidden-state ring (same cell: same ring geometry)
        if rv is None or rvh[1] is not rv[1] or rvh[0].shape[1] != n_agents * 64:
            raise RuntimeError("fused_actor_forward: hidden states are read in place only together with the observations of the same window")
        hid_slab_stride = rvh[0].shape[1]
        hidden = rvh[0]
    means = (out or {}).get("means")
    if means is None:
        means = th.empty(rows, a.action_dim, dtype=th.float32, device=obs.device)
    hid_out = (out or {}).get("hidden_out")
    if hid_out is None:
        hid_out = th.empty(rows, 64, dtype=th.float32, device=obs.device)
    args = _lib.FlexActorArgs()
    args.rows, args.n_agents, args.obs_dim, args.act_dim = rows, n_agents, obs_dim, a.action_dim
    args.agent_id, args.layernorm, args.ln_eps = int(bool(agent_id)), int(bool(a.layernorm)), 1e-5
    args.variant = int(ACTOR_VARIANT if variant is None else variant)
    action = env_action = None
    explore = noise is not None or rng_state is not None
    if noise is not None:
        noise = noise.reshape(rows, a.action_dim).to(th.float32).contiguous()
    elif rng_state is not None:
        if not (rng_state.is_cuda and rng_state.dtype == th.int64 and rng_state.numel() == 2 and rng_state.is_contiguous()):
            raise ValueError("rng_state must be a contiguous int64 device tensor [seed, step]")
        args.rng_state = rng_state.data_ptr()
    if ring_cursor is not None:
        args.cursor, args.obs_slab_stride, args.hid_slab_stride = ring_cursor.data_ptr(), int(obs_slab_stride), int(hid_slab_stride)
        if cursor_out is not None:
            args.cursor_out = cursor_out.data_ptr()
    if explore:
        action, env_action = (out or {}).get("action"), (out or {}).get("env_action")
        if env_action is None:
            env_action = th.empty_like(means)
        if action is None:
            action = th.empty_like(means)
        args.std, args.action_low, args.action_high = float(std), float(low), float(high)
    ln = agent.layernorm if a.layernorm else None
    if ln is not None:
        args.ln_eps = float(ln.eps)
    for name, t in (("obs", obs), ("hidden_in", hidden), ("fc1_w", agent.fc1.weight), ("fc1_b", agent.fc1.bias),
                    ("ln_w", ln.weight if ln is not None else None), ("ln_b", ln.bias if ln is not None else None),
                    ("w_ih", agent.rnn.weight_ih), ("w_hh", agent.rnn.weight_hh), ("b_ih", agent.rnn.bias_ih),
                    ("b_hh", agent.rnn.bias_hh), ("fc2_w", agent.fc2.weight), ("fc2_b", agent.fc2.bias),
                    ("means", means), ("hidden_out", hid_out), ("noise", noise), ("action", action),
                    ("env_action", env_action)):
        if t is not None and not t.is_contiguous():
            note_fallback("actor_forward", f"non-contiguous {name}")
            return None
        setattr(args, name, None if t is None else t.data_ptr())
    args.ring_slabs = int(ring_slabs)
    if obs_source is not None:
        args.obs, args.obs_pushed = obs_source.ring, obs_source.pushed
        args.obs_row_stride, args.obs_pushed_stride = obs_source.row_stride, obs_source.pushed_stride
        args.obs_slots, args.obs_slot_w = obs_source.slots, obs_source.slot_w
    if launch is not None:
        launch(args)
        return (means, hid_out, action, env_action) if explore else (means, hid_out)
    rc = lib.flexnet_actor_forward(C.byref(args), C.c_void_p(th.cuda.current_stream().cuda_stream))
    if rc == _lib.FLEXNET_EUNSUPPORTED:
        note_fallback("actor_forward", "FLEXNET_EUNSUPPORTED from flexnet_actor_forward")
        return None
    _lib.check(rc, "flexnet_actor_forward")
    return (means, hid_out, action, env_action) if explore else (means, hid_out)


_WGRAD_WS = {}
WGRAD_MIN_ROWS = 2048          # below this the library GEMM is launch-bound either way


# the finish of the fused value loss rides in the first-layer weight gradient's second-stage launch (FLEX_WGRAD_FINISH_RIDER=0:
# two launches, the A/B switch)
WGRAD_FINISH_RIDER = os.environ.get("FLEX_WGRAD_FINISH_RIDER", "1") != "0"


def tall_wgrad_supported(dy, x):
    return (dy.is_cuda and x.is_cuda and dy.dtype == th.float32 and x.dtype == th.float32 and dy.dim() == 2
            and x.dim() == 2 and dy.shape[0] == x.shape[0] and 1 <= dy.shape[1] <= 192 and x.shape[1] >= 1
            and (dy.shape[0] <= 1 or (dy.stride(1) == 1 and x.stride(1) == 1 and dy.stride(0) >= dy.shape[1]
                                      and x.stride(0) >= x.shape[1] and max(dy.stride(0), x.stride(0)) < (1 << 24))))


def tall_wgrad(dy, x, out=None, accumulate=False, colsum=None, x2=None, out2=None, critic_finish=None):
    """dW[m, n] = sum_k dy[k, m] * x[k, n] (csrc/wgrad.hip: include/flexnet.h flexnet_wgrad) — the weight gradient of
    y = x @ W.T over a tall batch.  Row-strided views (column slices of the packed replay rows) are read in place.
    The workspace is per device: calls are expected on one stream at a time (the update's).
    ``x2`` / ``out2``: a second input block of the same layer (out2 = dy.T @ x2) — in the same launch where the kernel
    takes [x | x2] as one operand (64 output rows, x wider than 64 columns and a multiple of 5), else a second call.
    ``critic_finish`` = (FlexCriticTailArgs, FlexTdLossArgs): phase 2 of flexnet_critic_td_backward_phases rides in this
    call's second-stage launch (flexnet_wgrad_critic_finish) — or is launched on its own first where the rider does not apply."""
    import ctypes as C
    from . import _lib
    lib = _lib.load()
    k, m = dy.shape
    n = x.shape[1]
    if x2 is not None and not (32 < m <= 64 and n > 64 and n % 5 == 0 and k > 1 and tall_wgrad_supported(dy, x2)
                               and out2.stride(1) == 1):
        tall_wgrad(dy, x2, out=out2, accumulate=accumulate)
        x2 = None
    if k <= 1:
        dy, x = dy.contiguous(), x.contiguous()
    if out is None:
        out = th.empty(m, n, dtype=th.float32, device=dy.device)
    elif out.shape != (m, n) or out.dtype != th.float32 or (m > 1 and (out.stride(1) != 1 or out.stride(0) < n)):
        raise ValueError("tall_wgrad: `out` must be an fp32 [m, n] tensor or a column block of a wider one")
    if dy.device not in _WGRAD_WS:
        _WGRAD_WS[dy.device] = th.empty(_lib.FLEXNET_WGRAD_WS_FLOATS, dtype=th.float32, device=dy.device)
    ws = _WGRAD_WS[dy.device]
    a = _lib.FlexWgradArgs()
    a.k, a.m, a.n = k, m, n
    a.lda, a.ldb = (dy.stride(0), x.stride(0)) if k > 1 else (m, n)
    a.a, a.b, a.c = dy.data_ptr(), x.data_ptr(), out.data_ptr()
    rv = ring_view_of(x)
    if rv is not None:           # in place from the replay's stacked-observation ring (RING_VIEWS)
        if x.stride(0) != rv[0].shape[1] or k <= 1:
            raise RuntimeError("tall_wgrad: a placeholder of the stacked-observation ring must be read as whole rows")
        a.b, a.b_row_cell = rv[0].data_ptr(), rv[1].data_ptr()
    a.workspace, a.workspace_floats, a.accumulate = ws.data_ptr(), ws.numel(), int(accumulate)
    a.ldc = out.stride(0) if m > 1 else n
    if colsum is not None:          # [m] <- sum_k dy[k, :], the bias gradient, from the same pass over dy
        a.colsum = colsum.data_ptr()
    if x2 is not None:
        a.b2, a.c2, a.ldb2, a.n2, a.ldc2 = x2.data_ptr(), out2.data_ptr(), x2.stride(0), x2.shape[1], out2.stride(0)
    stream = C.c_void_p(th.cuda.current_stream().cuda_stream)
    if critic_finish is not None:
        cargs, targs = critic_finish
        rc = lib.flexnet_wgrad_critic_finish(C.byref(a), C.byref(cargs), C.byref(targs), stream) if WGRAD_FINISH_RIDER else _lib.FLEXNET_EUNSUPPORTED
        if rc == 0:
            return out
        if rc != _lib.FLEXNET_EUNSUPPORTED:
            _lib.check(rc, "flexnet_wgrad_critic_finish")
        _lib.check(lib.flexnet_critic_td_backward_phases(C.byref(cargs), C.byref(targs), 2, stream), "flexnet_critic_td_backward")
    _lib.check(lib.flexnet_wgrad(C.byref(a), stream), "flexnet_wgrad")
    return out


class _TallLinear(th.autograd.Function):
    """y = x @ W.T (+ b) for a tall x: the library GEMM forward and for dx, csrc/wgrad.hip for dW."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        return th.addmm(b, x, w.t()) if b is not None else x @ w.t()

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dx = dw = db = None
        want_db = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            if tall_wgrad_supported(dy, x):
                if want_db:
                    db = th.empty(dy.shape[1], dtype=th.float32, device=dy.device)
                dw = tall_wgrad(dy, x, colsum=db)
            else:
                dw = dy.t() @ x
        if ctx.needs_input_grad[0]:
            dx = dy @ w
        if want_db and db is None:
            db = dy.sum(0)
        return dx, dw, db


def tall_linear(x, w, b=None):
    """F.linear for [rows, in] inputs; rows >= WGRAD_MIN_ROWS on the GPU take the hand-written weight gradient."""
    if x.is_cuda and x.dim() == 2 and x.shape[0] >= WGRAD_MIN_ROWS and w.shape[0] <= 192 and x.dtype == th.float32:
        return _TallLinear.apply(x, w, b)
    return F.linear(x, w, b)


# In-place windows of the replay's stacked-observation ring (replay_buffer.DeviceReplayBuffer.enable_stacked_ring; round 5).
# A captured sub-update needs static addresses, so until round 4 every sub-update began by gathering its window's stacked
# observations into a static batch (flexnet_gather_window: 23 us at 32 768 samples, 79 us at 131 072 — a tenth of an update event
# at the reference's sample reuse, each observation expanded ~6 times).  With the ring kept stacked (every slab expanded ONCE,
# when it enters the replay) the window is a contiguous run of ring rows, and the kernels that read observations take the ring's
# base plus a DEVICE cell holding the window's first row (FlexLinear2Args.x1_row_cell, FlexWgradArgs.b_row_cell): the graph is
# static, the window moves.  The static batch's `state` tensor is then a placeholder full of NaN, registered here by its
# data pointer: a consumer that does not know the ring reads NaN, not yesterday's observations.
RING_VIEWS = {}              # placeholder data_ptr -> (ring tensor [rows, width], device int64 cell)


def register_ring_view(placeholder, ring, cell):
    """``placeholder`` (a NaN-filled [rows, width] tensor that stands where a gathered window used to be) is read as rows
    cell[0] .. of ``ring`` by the kernels that know how.  Entries whose placeholder has been freed are dropped first: they
    hold a view of their ring, and a dropped trainer's 2.9 GB ring must not outlive it."""
    import weakref
    for key in [k for k, v in RING_VIEWS.items() if v[2]() is None]:
        del RING_VIEWS[key]
    RING_VIEWS[placeholder.data_ptr()] = (ring, cell, weakref.ref(placeholder))


def ring_view_of(t):
    """(ring, cell) if ``t`` is (a whole-row view of) a registered placeholder, else None."""
    if not RING_VIEWS or not t.is_cuda:
        return None
    key = t.data_ptr()
    rv = RING_VIEWS.get(key)
    if rv is None:
        return None
    if rv[2]() is None:          # the placeholder is gone (its trainer was dropped): the address may be anybody's now
        del RING_VIEWS[key]
        return None
    return rv


LINEAR2_MIN_ROWS = 8192      # below: the library call (a launch of this kernel loads 189 KB of weights into every CU's registers)
LINEAR2_MAX_ROWS = 65536     # above: the library pair again — measured (tools/linear_bench.py, profiles/r05f_linear_bench.txt): 39.6 vs
#                              40-47 us at 32 768 rows, 41.4 vs 71.4 at 36 864, but 138 vs 128 at 131 072 (both within 10 % of what the
#                              fp32 matrix pipe sustains at the clock it holds under this load; the kernel's launch cost is lower, its slope
#                              slightly higher)
CRITIC_FC1_FUSED = True      # (tests switch it off to compare with the two library GEMMs)


def critic_first_layer(bias, obs2d, act2d, W, c_act):
    """``bias + obs2d @ W[:, :no].T + act2d @ W[:, c_act:c_act + na].T`` — the shared part of the centralised critic's
    first layer (mlp_critic.py:25-26 on maddpg.py:33-54's input).  On the GPU at update sizes: ONE launch of
    csrc/linear.hip (include/flexnet.h: flexnet_linear2 — weights stationary in registers, exact fp32 on the matrix cores, no
    second pass over the [b, 64] result; round 5); otherwise the two library GEMMs of rounds 1-4."""
    no, na_ = obs2d.shape[1], act2d.shape[1]
    rv = ring_view_of(obs2d)
    if rv is not None and (th.is_grad_enabled() or obs2d.stride(0) != no or rv[0].shape[1] != no):
        raise RuntimeError("critic_first_layer: a placeholder of the stacked-observation ring reached a path that cannot read it in place")
    ok = (CRITIC_FC1_FUSED and obs2d.is_cuda and obs2d.dtype == th.float32 and act2d.dtype == th.float32 and W.dtype == th.float32
          and (rv is not None or LINEAR2_MIN_ROWS <= obs2d.shape[0] <= LINEAR2_MAX_ROWS) and W.shape[0] == 64 and no % 8 == 0 and na_ % 4 == 0
          and obs2d.stride(0) % 4 == 0 and act2d.stride(0) % 4 == 0
          and obs2d.stride(1) == 1 and act2d.stride(1) == 1 and W.stride(1) == 1 and bias.is_contiguous()
          and not th.is_grad_enabled())
    if ok:
        import ctypes as C
        from . import _lib
        out = th.empty(obs2d.shape[0], 64, dtype=th.float32, device=obs2d.device)
        a = _lib.FlexLinear2Args()
        a.rows, a.k1, a.k2 = obs2d.shape[0], no, na_
        a.ld1, a.ld2, a.ldw, a.c1, a.c2 = obs2d.stride(0), act2d.stride(0), W.stride(0), 0, int(c_act)
        a.x1, a.x2, a.w, a.bias, a.out = obs2d.data_ptr(), act2d.data_ptr(), W.data_ptr(), bias.data_ptr(), out.data_ptr()
        if rv is not None:
            a.x1, a.x1_row_cell = rv[0].data_ptr(), rv[1].data_ptr()
        rc = _lib.load().flexnet_linear2(C.byref(a), C.c_void_p(th.cuda.current_stream().cuda_stream))
        if rc == 0:
            return out
        if rv is not None:
            raise RuntimeError(f"flexnet_linear2 declined an in-place window of the stacked-observation ring (code {rc})")
        if rc != _lib.FLEXNET_EUNSUPPORTED:
            _lib.check(rc, "flexnet_linear2")
        note_fallback("critic_fc1", "FLEXNET_EUNSUPPORTED from flexnet_linear2")
    if rv is not None:
        raise RuntimeError("critic_first_layer: an in-place window of the stacked-observation ring needs the fused kernel")
    shared = th.addmm(bias, obs2d, W[:, :no].t())
    shared.addmm_(act2d, W[:, c_act:c_act + na_].t())
    return shared


_TD_WS = {}


# Cross-rank reward statistics ("or all-reduce 2 x 5 moments"; off unless the trainer switches it on):
# model.py:321-322 normalises the reward with BATCH statistics; with data-parallel ranks each rank's batch is a shard, and
# the full-batch statistics are the sums over the ranks of the per-rank sums.  GPU path: the statistics pass of
# csrc/tdloss.hip leaves its per-block partial sums in the workspace, ONE all-reduce of those 8 KB makes them global
# (flexnet_td_stats -> all-reduce -> the call with stats_ready / stat_rows); CPU path: sync_batchnorm below.
# The switch lives on the BatchNorm MODULE (``bn.flex_sync_ranks``, set by the trainer that owns the model): a second
# trainer in the same process (bench legs, an evaluation trainer, tests) cannot change the behaviour of the first one,
# and a trainer's graphed and eager paths read the same flag (it used to be a module global).


def set_reward_bn_sync(bn, on):
    if bn is not None:
        object.__setattr__(bn, "flex_sync_ranks", bool(on))


def _sync_active(bn):
    import torch.distributed as dist
    return (bn is not None and getattr(bn, "flex_sync_ranks", False) and dist.is_available() and dist.is_initialized()
            and dist.get_world_size() > 1)


# Reward statistics filed ahead of the loss (round 5).  The trainer's captured value sub-update refreshes its static batch with
# ONE gather launch per window; the statistics pass of the fused value loss needs the batch's rewards only, so it rides in that
# launch (flexnet_gather_rows_td) instead of being the graph's own 6.5-us launch between the first-layer product and the
# backward kernel.  Protocol: the trainer OFFERS the static reward tensor (offer_td_stats) before the warm-up steps;
# _CriticTdLossFn.forward, meeting that tensor, takes the offer — stats_ready = 1, no statistics launch — and records that it
# did; the trainer then keeps the rider for every refresh of that batch (TransReplayBuffer.gather(td=...)) or withdraws it.
TD_OFFERS = {}              # data_ptr of a static [rows, n] reward tensor -> {"taken": bool}


def offer_td_stats(reward):
    """Offer to file the reward statistics of ``reward`` (a static batch tensor) with every refresh of it.  Returns
    (record, FlexTdLossArgs for the rider); record["taken"] says after a loss has run whether the fused value loss used them."""
    from . import _lib
    import weakref
    r2 = reward.reshape(reward.shape[0], -1)
    rec = {"taken": False, "ref": weakref.ref(reward)}        # (a dead tensor's address may be handed out again: checked on use)
    TD_OFFERS[r2.data_ptr()] = rec
    rows, n = r2.shape
    if r2.device not in _TD_WS:
        _TD_WS[r2.device] = th.empty(_lib.FLEXNET_TD_WS_FLOATS // 2, dtype=th.float64, device=r2.device)
    ws = _TD_WS[r2.device]
    a = _lib.FlexTdLossArgs()
    a.rows, a.n_agents, a.normalise, a.reward = rows, n, 1, r2.data_ptr()
    a.workspace, a.workspace_floats = ws.data_ptr(), 2 * ws.numel()
    return rec, a


def withdraw_td_stats(reward):
    TD_OFFERS.pop(reward.reshape(reward.shape[0], -1).data_ptr(), None)


def _td_sync_stats(a, ws, bn):
    """If cross-rank statistics are on for ``bn``: statistics pass, all-reduce of the partial sums, and mark ``a``
    (FlexTdLossArgs) so that the call that follows uses them instead of computing its own.  ``ws``: the fp64 workspace
    tensor of ``a``."""
    if not (a.normalise and _sync_active(bn)):
        return
    import ctypes as C
    import torch.distributed as dist
    from . import _lib
    _lib.check(_lib.load().flexnet_td_stats(C.byref(a), C.c_void_p(th.cuda.current_stream().cuda_stream)), "flexnet_td_stats")
    dist.all_reduce(ws[:_lib.FLEXNET_TD_STAT_DOUBLES], op=dist.ReduceOp.SUM)
    a.stats_ready, a.stat_rows = 1, int(a.rows) * dist.get_world_size()


def sync_batchnorm(bn, x):
    """``bn(x)`` of a training-mode nn.BatchNorm1d on [rows, n] with the batch statistics taken over ALL ranks' rows (equal
    shards): normalised output, running statistics and num_batches_tracked moved as the module moves them.  Plain tensor
    ops (the eager / CPU path); without an initialised process group of more than one rank it is ``bn(x)``."""
    if not (_sync_active(bn) and bn.training):
        return bn(x)
    import torch.distributed as dist
    xd = x.double()
    mom = th.cat([xd.sum(0), (xd * xd).sum(0)])
    dist.all_reduce(mom, op=dist.ReduceOp.SUM)
    rows = x.shape[0] * dist.get_world_size()
    n = x.shape[1]
    mean = mom[:n] / rows
    var = (mom[n:] / rows - mean * mean).clamp_min(0.0)
    out = (xd - mean) / th.sqrt(var + bn.eps)
    if bn.affine:
        out = out * bn.weight.double() + bn.bias.double()
    if bn.track_running_stats:
        with th.no_grad():
            m = bn.momentum
            unbiased = var * rows / (rows - 1) if rows > 1 else var
            bn.running_mean.mul_(1 - m).add_((m * mean).to(bn.running_mean.dtype))
            bn.running_var.mul_(1 - m).add_((m * unbiased).to(bn.running_var.dtype))
            bn.num_batches_tracked += 1
    return out.to(x.dtype)


class _TdLossFn(th.autograd.Function):
    """mean((BatchNorm(reward) + gamma (1 - done) next_q - q)^2) (maddpg.py:100-123 behind model.py:308-323) with its
    gradient w.r.t. q, in three small launches (csrc/tdloss.hip); the BatchNorm module's running statistics are
    updated in place as its own forward would."""

    @staticmethod
    def forward(ctx, q, next_q, reward, done, gamma, bn, update_stats=True):
        import ctypes as C
        from . import _lib
        lib = _lib.load()
        rows, n = reward.shape
        q2, nq, r, d = q.reshape(rows, n).contiguous(), next_q.reshape(rows, n).contiguous(), reward.contiguous(), \
            done.reshape(rows).contiguous()
        dq = th.empty_like(q2)
        loss = th.empty((), dtype=th.float32, device=q.device)
        if q.device not in _TD_WS:
            _TD_WS[q.device] = th.empty(_lib.FLEXNET_TD_WS_FLOATS // 2, dtype=th.float64, device=q.device)
        ws = _TD_WS[q.device]
        a = _lib.FlexTdLossArgs()
        a.rows, a.n_agents, a.normalise, a.gamma = rows, n, int(bn is not None), float(gamma)
        a.reward, a.done, a.next_q, a.q = r.data_ptr(), d.data_ptr(), nq.data_ptr(), q2.data_ptr()
        if bn is not None:
            a.bn_eps, a.bn_momentum = float(bn.eps), float(bn.momentum)
            if bn.affine:
                a.bn_weight, a.bn_bias = bn.weight.data_ptr(), bn.bias.data_ptr()
            if bn.track_running_stats and update_stats:
                a.running_mean, a.running_var = bn.running_mean.data_ptr(), bn.running_var.data_ptr()
                a.num_batches_tracked = bn.num_batches_tracked.data_ptr()
        a.dq, a.loss = dq.data_ptr(), loss.data_ptr()
        a.workspace, a.workspace_floats = ws.data_ptr(), 2 * ws.numel()
        _td_sync_stats(a, ws, bn)
        _lib.check(lib.flexnet_td_loss(C.byref(a), C.c_void_p(th.cuda.current_stream().cuda_stream)), "flexnet_td_loss")
        ctx.save_for_backward(dq)
        ctx.q_shape = q.shape
        return loss

    @staticmethod
    def backward(ctx, grad_loss):
        (dq,) = ctx.saved_tensors
        from .util import is_unit_seed
        if is_unit_seed(grad_loss):                   # the trainer's root gradient: dq is the answer as it stands
            return dq.view(ctx.q_shape), None, None, None, None, None, None
        return (dq * grad_loss).view(ctx.q_shape), None, None, None, None, None, None


def td_loss_supported(q, next_q, reward, done, bn):
    """The fused value loss covers fp32 GPU tensors, up to 8 agents, and a BatchNorm1d in training mode with momentum
    (or no normalisation at all)."""
    ok = (q.is_cuda and q.dtype == th.float32 and next_q.dtype == th.float32 and reward.dtype == th.float32
          and done.dtype == th.float32 and reward.dim() == 2 and 1 <= reward.shape[1] <= 8 and reward.shape[0] >= 1
          and q.numel() == reward.numel() and next_q.numel() == reward.numel() and done.numel() == reward.shape[0]
          and not next_q.requires_grad)
    if bn is not None:
        ok = ok and bn.training and bn.momentum is not None and isinstance(bn, nn.BatchNorm1d) \
            and bn.num_features == reward.shape[1] and (not bn.track_running_stats or bn.running_mean.dtype == th.float32)
    return bool(ok)


def batchnorm_update_running_stats(bn, x):
    """What a training-mode ``bn(x)`` does to ``running_mean`` / ``running_var`` / ``num_batches_tracked``, without the
    normalised output (csrc/tdloss.hip statistics pass): for get_loss calls that normalise the reward only for the
    module's bookkeeping (the policy loss never reads it, model.py:308-323)."""
    import ctypes as C
    from . import _lib
    lib = _lib.load()
    rows, n = x.shape
    x = x.contiguous()
    if x.device not in _TD_WS:
        _TD_WS[x.device] = th.empty(_lib.FLEXNET_TD_WS_FLOATS // 2, dtype=th.float64, device=x.device)
    ws = _TD_WS[x.device]
    a = _lib.FlexTdLossArgs()
    a.rows, a.n_agents, a.normalise = rows, n, 1
    a.bn_eps, a.bn_momentum = float(bn.eps), float(bn.momentum)
    a.reward = x.data_ptr()
    a.running_mean, a.running_var = bn.running_mean.data_ptr(), bn.running_var.data_ptr()
    a.num_batches_tracked = bn.num_batches_tracked.data_ptr()
    a.workspace, a.workspace_floats = ws.data_ptr(), 2 * ws.numel()
    _td_sync_stats(a, ws, bn)
    _lib.check(lib.flexnet_td_loss(C.byref(a), C.c_void_p(th.cuda.current_stream().cuda_stream)), "flexnet_td_loss")


def batchnorm_stats_supported(bn, x):
    return bool(x.is_cuda and x.dtype == th.float32 and x.dim() == 2 and 1 <= x.shape[1] <= 8 and x.shape[0] >= 1
                and isinstance(bn, nn.BatchNorm1d) and bn.training and bn.momentum is not None and bn.track_running_stats
                and bn.num_features == x.shape[1] and bn.running_mean.dtype == th.float32)


def td_loss(q, next_q, reward, done, gamma, bn=None, update_stats=True):
    """``update_stats=False``: the same batch-statistics normalisation without moving the module's running statistics
    (a second loss term on the same normalised reward, matd3.py:141-148)."""
    return _TdLossFn.apply(q, next_q, reward, done, gamma, bn, update_stats)


_LNRELU_WS = {}


def _lnrelu_args(z, bias, id_cols, ln_w, ln_b, eps, n_agents):
    from . import _lib
    a = _lib.FlexLnReluArgs()
    a.rows, a.n_agents, a.layernorm, a.ln_eps = z.shape[0], n_agents, int(ln_w is not None), float(eps)
    a.z = z.data_ptr()
    a.bias = None if bias is None else bias.data_ptr()
    a.id_cols = None if id_cols is None else id_cols.data_ptr()
    if ln_w is not None:
        a.ln_w, a.ln_b = ln_w.data_ptr(), ln_b.data_ptr()
    return a


class _LnReluFn(th.autograd.Function):
    """relu(LayerNorm(z + bias + id_cols[r % n])) (rnn_agent.py:25-29 after the fc1 GEMM) with a hand-written backward
    (csrc/lnrelu.hip): dz and the gradients of bias, id columns and the LayerNorm pair in two launches."""

    @staticmethod
    def forward(ctx, z, bias, id_cols, ln_w, ln_b, eps, n_agents):
        import ctypes as C
        from . import _lib
        lib = _lib.load()
        z = z.contiguous()
        id_cols = None if id_cols is None else id_cols.contiguous()
        out = th.empty_like(z)
        a = _lnrelu_args(z, bias, id_cols, ln_w, ln_b, eps, n_agents)
        a.out = out.data_ptr()
        _lib.check(lib.flexnet_lnrelu_forward(C.byref(a), C.c_void_p(th.cuda.current_stream().cuda_stream)),
                   "flexnet_lnrelu_forward")
        ctx.eps, ctx.n_agents = eps, n_agents
        ctx.save_for_backward(z, bias, id_cols, ln_w, ln_b)
        return out

    @staticmethod
    def backward(ctx, dout):
        import ctypes as C
        from . import _lib
        lib = _lib.load()
        z, bias, id_cols, ln_w, ln_b = ctx.saved_tensors
        dout = dout.contiguous()
        dz = th.empty_like(z)
        small = th.empty(3 + ctx.n_agents, 64, dtype=th.float32, device=z.device)
        a = _lnrelu_args(z, bias, id_cols, ln_w, ln_b, ctx.eps, ctx.n_agents)
        a.dout, a.dz = dout.data_ptr(), dz.data_ptr()
        if ln_w is not None:
            a.d_ln_w, a.d_ln_b = small[0].data_ptr(), small[1].data_ptr()
        if bias is not None:
            a.d_bias = small[2].data_ptr()
        if id_cols is not None:
            a.d_id = small[3:].data_ptr()
        if z.device not in _LNRELU_WS:
            _LNRELU_WS[z.device] = th.empty(_lib.FLEXNET_LNRELU_WS_FLOATS, dtype=th.float32, device=z.device)
        ws = _LNRELU_WS[z.device]
        a.workspace, a.workspace_floats = ws.data_ptr(), ws.numel()
        _lib.check(lib.flexnet_lnrelu_backward(C.byref(a), C.c_void_p(th.cuda.current_stream().cuda_stream)),
                   "flexnet_lnrelu_backward")
        return (dz, None if bias is None else small[2], None if id_cols is None else small[3:],
                None if ln_w is None else small[0], None if ln_w is None else small[1], None, None)


# the first layer's backward inside the gate-gradient launch (csrc/gru.hip, gru_backward_fused_kernel); False: the round-2
# composition (pointwise gate gradients, library GEMM for dx, csrc/lnrelu.hip) — kept as the cross-check of tests/test_gru_gpu.py
GRU_BWD_FUSED = True
_DEBUG_KEEP = None                # tools/gru_diag.py: a dict that receives the backward's dz


class _ActorTrainFn(th.autograd.Function):
    """The whole actor of rnn_agent.py:25-33 for an update batch as ONE autograd node: the forward is the fused
    matrix-core kernel of csrc/actor.hip (fc1 -> LayerNorm -> ReLU -> GRUCell -> fc2 chained through registers) with its
    `save_*` outputs, the backward is csrc/gru.hip (gate gradients, fc2's input gradient folded in), csrc/wgrad.hip for
    every weight / bias gradient, one library GEMM for dx and csrc/lnrelu.hip for the first layer's epilogue.  Replaces
    the composition fc1 GEMM + lnrelu + two gate GEMMs + ATen's pointwise GRU cell + fc2 GEMM (309 us forward at 163 840
    rows) and their autograd backward.  Observations and the previous hidden state are replayed tensors: no gradient."""

    @staticmethod
    def forward(ctx, obs, hidden, n_agents, agent_id, fc1_w, fc1_b, ln_w, ln_b, ln_eps, w_ih, w_hh, b_ih, b_hh, fc2_w, fc2_b):
        import ctypes as C
        from . import _lib
        lib = _lib.load()
        rows, o = obs.shape
        act_dim = fc2_w.shape[0]
        obs, hidden = obs.contiguous(), hidden.contiguous()
        means = th.empty(rows, act_dim, dtype=th.float32, device=obs.device)
        hid_out = th.empty(rows, 64, dtype=th.float32, device=obs.device)
        saved = th.empty(6, rows, 64, dtype=th.float32, device=obs.device)          # z1 | x | r | z | n | hn
        a = _lib.FlexActorArgs()
        a.rows, a.n_agents, a.obs_dim, a.act_dim = rows, n_agents, o, act_dim
        a.agent_id, a.layernorm, a.ln_eps, a.variant = int(bool(agent_id)), int(ln_w is not None), float(ln_eps), 0
        for name, t in (("obs", obs), ("hidden_in", hidden), ("fc1_w", fc1_w), ("fc1_b", fc1_b), ("ln_w", ln_w), ("ln_b", ln_b),
                        ("w_ih", w_ih), ("w_hh", w_hh), ("b_ih", b_ih), ("b_hh", b_hh), ("fc2_w", fc2_w), ("fc2_b", fc2_b),
                        ("means", means), ("hidden_out", hid_out), ("save_z1", saved[0]), ("save_x", saved[1]),
                        ("save_r", saved[2]), ("save_z", saved[3]), ("save_n", saved[4]), ("save_hn", saved[5])):
            if t is not None:
                if not t.is_contiguous():
                    raise ValueError(f"actor training forward: {name} must be contiguous")
                setattr(a, name, t.data_ptr())
        _lib.check(lib.flexnet_actor_forward(C.byref(a), C.c_void_p(th.cuda.current_stream().cuda_stream)),
                   "flexnet_actor_forward")
        ctx.n_agents, ctx.agent_id, ctx.ln_eps = n_agents, bool(agent_id), float(ln_eps)
        ctx.save_for_backward(obs, hidden, hid_out, saved, fc1_w, fc1_b, ln_w, ln_b, w_ih, fc2_w)
        ctx.mark_non_differentiable(hid_out)          # the new hidden state is returned for the caller's bookkeeping only
        ctx.set_materialize_grads(False)              # ... and must not cost a [rows, 64] zero fill per backward
        return means, hid_out

    @staticmethod
    def backward(ctx, d_means, _d_hid):
        import ctypes as C
        from . import _lib
        lib = _lib.load()
        obs, hidden, hid_out, saved, fc1_w, fc1_b, ln_w, ln_b, w_ih, fc2_w = ctx.saved_tensors
        rows, o = obs.shape
        n, dev = ctx.n_agents, obs.device
        stream = C.c_void_p(th.cuda.current_stream().cuda_stream)
        d_means = d_means.contiguous()
        d_gi = th.empty(rows, 192, dtype=th.float32, device=dev)
        d_gh = th.empty(rows, 192, dtype=th.float32, device=dev)
        g = _lib.FlexGruBwdArgs()
        g.rows, g.act_dim = rows, fc2_w.shape[0]
        g.d_means, g.fc2_w = d_means.data_ptr(), fc2_w.data_ptr()
        g.r, g.z, g.n, g.hn = saved[2].data_ptr(), saved[3].data_ptr(), saved[4].data_ptr(), saved[5].data_ptr()
        g.h_prev, g.d_gi, g.d_gh = hidden.data_ptr(), d_gi.data_ptr(), d_gh.data_ptr()
        fused = GRU_BWD_FUSED
        if fused:
            # ... and the first layer's backward in the same launch (round 3): dx = d_gi @ W_ih on the matrix cores, never
            # stored; LayerNorm / ReLU / bias / id columns backward; the id-column sums land in fc1's gradient through strides
            dz = th.empty(rows, 64, dtype=th.float32, device=dev)
            small = th.empty(3, 64, dtype=th.float32, device=dev)
            d_fc1_w = th.empty_like(fc1_w)
            if dev not in _LNRELU_WS:
                _LNRELU_WS[dev] = th.empty(_lib.FLEXNET_LNRELU_WS_FLOATS, dtype=th.float32, device=dev)
            ws = _LNRELU_WS[dev]
            g.w_ih, g.z1, g.fc1_w, g.dz = w_ih.data_ptr(), saved[0].data_ptr(), fc1_w.data_ptr(), dz.data_ptr()
            g.x = saved[1].data_ptr()                  # ReLU's mask: the forward's own output
            g.fc1_b = fc1_b.data_ptr() if fc1_b is not None else None
            g.fc1_ld, g.obs_dim, g.n_agents, g.agent_id = fc1_w.shape[1], o, n, int(ctx.agent_id)
            g.layernorm, g.ln_eps = int(ln_w is not None), ctx.ln_eps
            if ln_w is not None:
                g.ln_w, g.ln_b, g.d_ln_w, g.d_ln_b = ln_w.data_ptr(), ln_b.data_ptr(), small[0].data_ptr(), small[1].data_ptr()
            g.d_fc1_b = small[2].data_ptr()
            if ctx.agent_id:
                g.d_id = d_fc1_w.data_ptr() + 4 * o
                g.d_id_agent_stride, g.d_id_unit_stride = 1, fc1_w.shape[1]
            g.workspace, g.workspace_floats = ws.data_ptr(), ws.numel()
        _lib.check(lib.flexnet_gru_backward(C.byref(g), stream), "flexnet_gru_backward")
        # fc2: weight and bias gradients from one pass over d_means
        d_fc2_b = th.empty(fc2_w.shape[0], dtype=th.float32, device=dev)
        d_fc2_w = tall_wgrad(d_means, hid_out, colsum=d_fc2_b)
        # GRUCell: W_ih sees [dr | dz | dn], W_hh sees [dr | dz | dn r]; the biases are the column sums of the same passes
        d_b_ih = th.empty(192, dtype=th.float32, device=dev)
        d_w_ih = tall_wgrad(d_gi, saved[1], colsum=d_b_ih)
        d_b_hh = th.empty(192, dtype=th.float32, device=dev)
        d_w_hh = tall_wgrad(d_gh, hidden, colsum=d_b_hh)
        if fused:
            if _DEBUG_KEEP is not None:
                _DEBUG_KEEP["dz"] = dz
            tall_wgrad(dz, obs, out=d_fc1_w[:, :o])
            has_ln = ln_w is not None
            return (None, None, None, None, d_fc1_w, small[2], small[0] if has_ln else None, small[1] if has_ln else None, None,
                    d_w_ih, d_w_hh, d_b_ih, d_b_hh, d_fc2_w, d_fc2_b)
        # first layer: dx = d_gi @ W_ih, then the LayerNorm / ReLU / bias / id-column epilogue backward and fc1's weight
        dx = d_gi @ w_ih
        dz = th.empty(rows, 64, dtype=th.float32, device=dev)
        small = th.empty(3 + n, 64, dtype=th.float32, device=dev)
        id_cols = fc1_w[:, o:].t().contiguous() if ctx.agent_id else None
        la = _lnrelu_args(saved[0], fc1_b, id_cols, ln_w, ln_b, ctx.ln_eps, n)
        la.dout, la.dz = dx.data_ptr(), dz.data_ptr()
        if ln_w is not None:
            la.d_ln_w, la.d_ln_b = small[0].data_ptr(), small[1].data_ptr()
        la.d_bias = small[2].data_ptr()
        if id_cols is not None:
            la.d_id = small[3:].data_ptr()
        if dev not in _LNRELU_WS:
            _LNRELU_WS[dev] = th.empty(_lib.FLEXNET_LNRELU_WS_FLOATS, dtype=th.float32, device=dev)
        ws = _LNRELU_WS[dev]
        la.workspace, la.workspace_floats = ws.data_ptr(), ws.numel()
        _lib.check(lib.flexnet_lnrelu_backward(C.byref(la), stream), "flexnet_lnrelu_backward")
        if _DEBUG_KEEP is not None:
            _DEBUG_KEEP["dz"] = dz
        d_fc1_w = th.empty_like(fc1_w)
        tall_wgrad(dz, obs, out=d_fc1_w[:, :o])
        if ctx.agent_id:
            d_fc1_w[:, o:] = small[3:].t()
        has_ln = ln_w is not None
        return (None, None, None, None, d_fc1_w, small[2], small[0] if has_ln else None, small[1] if has_ln else None, None,
                d_w_ih, d_w_hh, d_b_ih, d_b_hh, d_fc2_w, d_fc2_b)


def actor_train_supported(agent, obs, n_agents, agent_id):
    """What the fused training pass covers: the RNN agent at 64 hidden units with ReLU, fp32 GPU rows >= WGRAD_MIN_ROWS."""
    a = agent.args
    W = agent.fc1.weight
    return (isinstance(agent, RNNAgent) and obs.is_cuda and obs.dtype == th.float32 and obs.dim() == 2
            and a.hid_size == 64 and a.hid_activation == "relu" and obs.shape[1] <= 144 and 1 <= n_agents <= 8
            and a.action_dim <= 8 and obs.shape[0] >= WGRAD_MIN_ROWS and obs.shape[0] % n_agents == 0
            and W.shape[1] == obs.shape[1] + (n_agents if agent_id else 0) and not obs.requires_grad
            and getattr(agent, "fused_training", True) and getattr(agent, "fused_epilogue", True))


def lnrelu_supported(agent, n_agents):
    """Configurations csrc/lnrelu.hip covers: 64 hidden units, ReLU, at most FLEXNET_MAX_AGENTS id columns."""
    a = agent.args
    return (a.hid_size == 64 and a.hid_activation == "relu" and 1 <= n_agents <= 8
            and getattr(agent, "fused_epilogue", True))


class _WideBatchLinear(th.autograd.Function):
    """y = x @ W.T for a tall x [B, K] (B ~ 1e4..1e5) and a small W [N, K], x without gradient.  The weight gradient
    dW = dy.T @ x has only N*K/tile output tiles (23 workgroups for 64 x 720) and a reduction over the whole batch:
    the library kernel leaves 90 % of the chip idle (250 us at batch 32 768).  Split-K by hand: the batch is cut into
    S slabs, one batched GEMM forms S partial dW, their sum is the gradient (S * 23 workgroups)."""

    SLABS = 16

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return x @ w.t()

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dw = dx = None
        if ctx.needs_input_grad[1]:
            b = x.shape[0]
            s = _WideBatchLinear.SLABS
            if b >= WGRAD_MIN_ROWS and tall_wgrad_supported(dy, x):
                dw = tall_wgrad(dy, x)
            elif b % s == 0 and b >= 4096:
                dw = th.bmm(dy.reshape(s, b // s, -1).transpose(1, 2), x.reshape(s, b // s, -1)).sum(0)
            else:
                dw = dy.t() @ x
        if ctx.needs_input_grad[0]:
            dx = dy @ w
        return dx, dw


def wide_batch_linear(x, w):
    return _WideBatchLinear.apply(x, w)


def critic_tail_supported(critic, x):
    a = critic.args
    ok = (x.is_cuda and x.dtype == th.float32 and x.dim() == 2 and x.shape[1] == 64 and a.hid_size == 64
          and a.hid_activation == "relu" and critic.fc3.out_features == 1 and getattr(critic, "fused_tail", True))
    if x.is_cuda and not ok and getattr(critic, "fused_tail", True):
        note_fallback("critic_tail", f"hid {a.hid_size}, act {a.hid_activation}, out {critic.fc3.out_features}, dtype {x.dtype}")
    return ok


CRITIC_VARIANT = 0              # 0: matrix-core forward / dz1-only backward (csrc/critic.hip); 1: the VALU kernels
CRITIC_PGRAD32 = 0              # matrix-core backward WITH parameter gradients: 0 = 16-row tiles, two wavefronts per SIMD;
#                                 1 = the 32-row kernel it replaced (tests and tools/critic_bench.py compare the two)


def _critic_args(z1, ln_w, ln_b, w2, b2, w3, b3, eps):
    from . import _lib
    a = _lib.FlexCriticTailArgs()
    a.rows, a.layernorm, a.ln_eps, a.variant = z1.shape[0], int(ln_w is not None), float(eps), CRITIC_VARIANT
    a.variant_pgrad32 = CRITIC_PGRAD32
    a.z1 = z1.data_ptr()
    if ln_w is not None:
        a.ln_w, a.ln_b = ln_w.data_ptr(), ln_b.data_ptr()
    a.fc2_w, a.fc2_b, a.fc3_w, a.fc3_b = w2.data_ptr(), b2.data_ptr(), w3.data_ptr(), b3.data_ptr()
    return a


_CRITIC_WS = {}


def _set_critic_ids(args, W, col0, n, dense=None):
    """The composed input's id-column table: ``dense`` [n, 64] if given, else columns col0 .. col0 + n - 1 of fc1.weight
    read where they are (the kernels stage the table in LDS through the two strides: no transposed copy per call)."""
    if dense is not None:
        args.z_id = dense.data_ptr()
    else:
        args.z_id = W[:, col0:col0 + n].data_ptr()
        args.z_id_agent_stride, args.z_id_unit_stride = 1, W.stride(0)


def _critic_workspace(device):
    """Per-device scratch for the backward kernel's per-block partial sums (18 MB, allocated once)."""
    from . import _lib
    if device not in _CRITIC_WS:
        _CRITIC_WS[device] = th.empty(_lib.FLEXNET_CRITIC_WS_FLOATS, dtype=th.float32, device=device)
    return _CRITIC_WS[device]


class _CriticTailFn(th.autograd.Function):
    """q = fc3(relu(fc2(relu(LayerNorm(z1))))) (mlp_critic.py:27-31) with a hand-written backward: gradients for z1 and
    for the six parameter tensors of the tail."""

    @staticmethod
    def forward(ctx, z1, ln_w, ln_b, w2, b2, w3, b3, eps):
        import ctypes as C
        from . import _lib
        lib = _lib.load()
        z1 = z1.contiguous()
        q = th.empty(z1.shape[0], 1, dtype=th.float32, device=z1.device)
        args = _critic_args(z1, ln_w, ln_b, w2, b2, w3, b3, eps)
        args.q = q.data_ptr()
        _lib.check(lib.flexnet_critic_tail_forward(C.byref(args), C.c_void_p(th.cuda.current_stream().cuda_stream)),
                   "flexnet_critic_tail_forward")
        ctx.eps = eps
        ctx.save_for_backward(z1, ln_w, ln_b, w2, b2, w3, b3)
        return q

    @staticmethod
    def backward(ctx, dq):
        import ctypes as C
        from . import _lib
        lib = _lib.load()
        z1, ln_w, ln_b, w2, b2, w3, b3 = ctx.saved_tensors
        dq = dq.contiguous()
        dz1 = th.empty_like(z1)
        args = _critic_args(z1, ln_w, ln_b, w2, b2, w3, b3, ctx.eps)
        args.dq, args.dz1 = dq.data_ptr(), dz1.data_ptr()
        if not any(ctx.needs_input_grad[1:7]):
            # the policy loss differentiates through a critic whose parameters take no step: dz1 only
            _lib.check(lib.flexnet_critic_tail_backward(C.byref(args), C.c_void_p(th.cuda.current_stream().cuda_stream)),
                       "flexnet_critic_tail_backward")
            return (dz1,) + (None,) * 7
        grads = th.zeros(64 * 64 + 64 * 4 + 1, dtype=th.float32, device=z1.device)      # one zero-fill for all six
        d_w2, d_b2, d_w3 = grads[:4096].view(64, 64), grads[4096:4160], grads[4160:4224].view(1, 64)
        d_g, d_b, d_b3 = grads[4224:4288], grads[4288:4352], grads[4352:4353]
        args.d_fc2_w, args.d_fc2_b, args.d_fc3_w, args.d_fc3_b = d_w2.data_ptr(), d_b2.data_ptr(), d_w3.data_ptr(), d_b3.data_ptr()
        if ln_w is not None:
            args.d_ln_w, args.d_ln_b = d_g.data_ptr(), d_b.data_ptr()
        ws = _critic_workspace(z1.device)
        args.workspace, args.workspace_floats = ws.data_ptr(), ws.numel()
        _lib.check(lib.flexnet_critic_tail_backward(C.byref(args), C.c_void_p(th.cuda.current_stream().cuda_stream)),
                   "flexnet_critic_tail_backward")
        has_ln = ln_w is not None
        return dz1, (d_g if has_ln else None), (d_b if has_ln else None), d_w2, d_b2, d_w3, d_b3, None


class _CriticTailComposedFn(th.autograd.Function):
    """The same tail on z1[b, i] = shared[b] + id_cols[i] (the critic's first-layer output when no agent's own-action
    gradient is needed, maddpg.py:38-54) without materialising the [b * n, 64] tensor on the way in."""

    @staticmethod
    def forward(ctx, shared, id_cols, ln_w, ln_b, w2, b2, w3, b3, eps):
        import ctypes as C
        from . import _lib
        lib = _lib.load()
        shared = shared.contiguous()
        n = id_cols.shape[0]
        rows = shared.shape[0] * n
        q = th.empty(rows, 1, dtype=th.float32, device=shared.device)
        args = _critic_args(shared, ln_w, ln_b, w2, b2, w3, b3, eps)
        args.rows, args.z1, args.z_shared, args.z_id, args.n_agents = rows, None, shared.data_ptr(), id_cols.data_ptr(), n
        if not id_cols.is_contiguous():          # fc1.weight's id columns, transposed view: read in place through strides
            args.z_id_agent_stride, args.z_id_unit_stride = id_cols.stride(0), id_cols.stride(1)
        args.q = q.data_ptr()
        _lib.check(lib.flexnet_critic_tail_forward(C.byref(args), C.c_void_p(th.cuda.current_stream().cuda_stream)),
                   "flexnet_critic_tail_forward")
        ctx.eps = eps
        ctx.save_for_backward(shared, id_cols, ln_w, ln_b, w2, b2, w3, b3)
        return q

    @staticmethod
    def backward(ctx, dq):
        import ctypes as C
        from . import _lib
        lib = _lib.load()
        shared, id_cols, ln_w, ln_b, w2, b2, w3, b3 = ctx.saved_tensors
        id_cols = id_cols.contiguous()
        n = id_cols.shape[0]
        rows = shared.shape[0] * n
        dq = dq.contiguous()
        dz1 = th.empty(rows, 64, dtype=th.float32, device=shared.device)
        grads = th.zeros(64 * 64 + 64 * 4 + 1, dtype=th.float32, device=shared.device)
        d_w2, d_b2, d_w3 = grads[:4096].view(64, 64), grads[4096:4160], grads[4160:4224].view(1, 64)
        d_g, d_b, d_b3 = grads[4224:4288], grads[4288:4352], grads[4352:4353]
        args = _critic_args(shared, ln_w, ln_b, w2, b2, w3, b3, ctx.eps)
        args.rows, args.z1, args.z_shared, args.z_id, args.n_agents = rows, None, shared.data_ptr(), id_cols.data_ptr(), n
        args.dq, args.dz1 = dq.data_ptr(), dz1.data_ptr()
        args.d_fc2_w, args.d_fc2_b, args.d_fc3_w, args.d_fc3_b = d_w2.data_ptr(), d_b2.data_ptr(), d_w3.data_ptr(), d_b3.data_ptr()
        if ln_w is not None:
            args.d_ln_w, args.d_ln_b = d_g.data_ptr(), d_b.data_ptr()
        ws = _critic_workspace(shared.device)
        args.workspace, args.workspace_floats = ws.data_ptr(), ws.numel()
        d_shared = th.empty_like(shared)
        d_id = th.empty(n, 64, dtype=th.float32, device=shared.device)
        args.d_z_shared, args.d_z_id = d_shared.data_ptr(), d_id.data_ptr()
        _lib.check(lib.flexnet_critic_tail_backward(C.byref(args), C.c_void_p(th.cuda.current_stream().cuda_stream)),
                   "flexnet_critic_tail_backward")
        has_ln = ln_w is not None
        return (d_shared, d_id, (d_g if has_ln else None), (d_b if has_ln else None), d_w2, d_b2, d_w3, d_b3, None)


class _CriticReplayedFn(th.autograd.Function):
    """The WHOLE shared-parameter critic on replayed (gradient-free) inputs, maddpg.py:33-76 + mlp_critic.py:25-33, as
    one autograd node: q[b, i] = tail(obs_all[b] W_obs^T + act_all[b] W_act^T + bias + W_id[:, i]).  Forward: two GEMMs
    (the second accumulates) and the composed tail kernel.  Backward: the tail backward kernel, two reductions of dz1,
    and fc1's weight gradient written block by block into ONE [64, in] tensor by csrc/wgrad.hip (its bias gradient is the
    column sum from the same pass) — instead of three column-slice gradients that autograd zero-fills, scatters and
    adds, plus a separate bias reduction."""

    @staticmethod
    def forward(ctx, obs2d, act2d, n_agents, W, bias, ln_w, ln_b, w2, b2, w3, b3, eps, twin=False):
        import ctypes as C
        from . import _lib
        lib = _lib.load()
        no, na_ = obs2d.shape[1], act2d.shape[1]
        shared = critic_first_layer(bias, obs2d, act2d, W, no + n_agents)
        # twin (matd3.py:64-67): the second head is the same network with the trailing 0/1 input flag set — fc1's last
        # column joins every agent's id column
        id_cols = (W[:, no:no + n_agents] + W[:, -1:]).t().contiguous() if twin else None
        ctx.twin = bool(twin)
        rows = shared.shape[0] * n_agents
        q = th.empty(rows, 1, dtype=th.float32, device=shared.device)
        args = _critic_args(shared, ln_w, ln_b, w2, b2, w3, b3, eps)
        args.rows, args.z1, args.z_shared, args.n_agents = rows, None, shared.data_ptr(), n_agents
        _set_critic_ids(args, W, no, n_agents, dense=id_cols)
        args.q = q.data_ptr()
        _lib.check(lib.flexnet_critic_tail_forward(C.byref(args), C.c_void_p(th.cuda.current_stream().cuda_stream)),
                   "flexnet_critic_tail_forward")
        ctx.eps, ctx.n_agents = eps, n_agents
        ctx.save_for_backward(obs2d, act2d, shared, id_cols, W, ln_w, ln_b, w2, b2, w3, b3)
        return q

    @staticmethod
    def backward(ctx, dq):
        import ctypes as C
        from . import _lib
        lib = _lib.load()
        obs2d, act2d, shared, id_cols, W, ln_w, ln_b, w2, b2, w3, b3 = ctx.saved_tensors      # (id_cols: twin only)
        n = ctx.n_agents
        no, na_ = obs2d.shape[1], act2d.shape[1]
        rows = shared.shape[0] * n
        dq = dq.contiguous()
        dz1 = th.empty(rows, 64, dtype=th.float32, device=shared.device)
        # the six tail gradients are STORED by the fixed-order second stage (overwrite_grads): no zero fill
        grads = th.empty(64 * 64 + 64 * 4 + 1, dtype=th.float32, device=shared.device)
        d_w2, d_b2, d_w3 = grads[:4096].view(64, 64), grads[4096:4160], grads[4160:4224].view(1, 64)
        d_g, d_b, d_b3 = grads[4224:4288], grads[4288:4352], grads[4352:4353]
        args = _critic_args(shared, ln_w, ln_b, w2, b2, w3, b3, ctx.eps)
        args.rows, args.z1, args.z_shared, args.n_agents = rows, None, shared.data_ptr(), n
        _set_critic_ids(args, W, no, n, dense=id_cols)
        args.dq, args.dz1 = dq.data_ptr(), dz1.data_ptr()
        args.d_fc2_w, args.d_fc2_b, args.d_fc3_w, args.d_fc3_b = d_w2.data_ptr(), d_b2.data_ptr(), d_w3.data_ptr(), d_b3.data_ptr()
        if ln_w is not None:
            args.d_ln_w, args.d_ln_b = d_g.data_ptr(), d_b.data_ptr()
        ws = _critic_workspace(shared.device)
        args.workspace, args.workspace_floats, args.overwrite_grads = ws.data_ptr(), ws.numel(), 1
        d_shared = th.empty_like(shared)                    # dz1 folded onto its two sources by the same call
        dW = th.empty_like(W)
        if ctx.twin:
            d_id = th.empty(n, 64, dtype=th.float32, device=shared.device)
        else:                                               # ... the id-column sums straight into dW[:, no:no + n]
            d_id = dW[:, no:no + n]
            args.d_z_id_agent_stride, args.d_z_id_unit_stride = 1, W.shape[1]
        args.d_z_shared, args.d_z_id = d_shared.data_ptr(), d_id.data_ptr()
        _lib.check(lib.flexnet_critic_tail_backward(C.byref(args), C.c_void_p(th.cuda.current_stream().cuda_stream)),
                   "flexnet_critic_tail_backward")
        d_bias = th.empty(64, dtype=th.float32, device=W.device)
        # fc1: the observation block and the action block of dW from one launch (the action block rides in the last column chunk)
        tall_wgrad(d_shared, obs2d, out=dW[:, :no], colsum=d_bias, x2=act2d, out2=dW[:, no + n:no + n + na_])
        if ctx.twin:
            dW[:, no:no + n] = d_id.t()
        if W.shape[1] > no + n + na_:
            dW[:, no + n + na_:] = 0.0
            if ctx.twin:                 # the flag column sees every agent's row: the sum of the id-column gradients
                tot = d_id[0]
                for i in range(1, n):
                    tot = tot + d_id[i]      # (n - 1 pointwise adds: no ATen reduction in a captured graph)
                dW[:, -1] = tot
        has_ln = ln_w is not None
        return (None, None, None, dW, d_bias, (d_g if has_ln else None), (d_b if has_ln else None), d_w2, d_b2, d_w3, d_b3,
                None, None)


class _CriticReplayedTwinFn(th.autograd.Function):
    """Both heads of MATD3's twin critic (matd3.py:33-86: ONE network, the second head with the trailing 0/1 input flag set)
    on replayed inputs as one node: fc1's output on [obs | act] is the same for both heads — only the id-column table
    differs (the flag column joins it) — so it is formed once, the tail runs forward and backward per head, and fc1's weight
    gradient is taken ONCE on the sum of the two heads' input gradients.  Returns cat([Q1, Q2]) as [2 rows, 1]."""

    @staticmethod
    def forward(ctx, obs2d, act2d, n_agents, W, bias, ln_w, ln_b, w2, b2, w3, b3, eps):
        import ctypes as C
        from . import _lib
        lib = _lib.load()
        stream = C.c_void_p(th.cuda.current_stream().cuda_stream)
        n = n_agents
        no, na_ = obs2d.shape[1], act2d.shape[1]
        shared = critic_first_layer(bias, obs2d, act2d, W, no + n)
        id2 = (W[:, no:no + n] + W[:, -1:]).t().contiguous()
        rows = shared.shape[0] * n
        q = th.empty(2 * rows, 1, dtype=th.float32, device=shared.device)
        for h in range(2):
            args = _critic_args(shared, ln_w, ln_b, w2, b2, w3, b3, eps)
            args.rows, args.z1, args.z_shared, args.n_agents = rows, None, shared.data_ptr(), n
            _set_critic_ids(args, W, no, n, dense=id2 if h else None)
            args.q = q[h * rows:].data_ptr()
            _lib.check(lib.flexnet_critic_tail_forward(C.byref(args), stream), "flexnet_critic_tail_forward")
        ctx.eps, ctx.n_agents = eps, n
        ctx.save_for_backward(obs2d, act2d, shared, id2, W, ln_w, ln_b, w2, b2, w3, b3)
        return q

    @staticmethod
    def backward(ctx, dq):
        import ctypes as C
        from . import _lib
        lib = _lib.load()
        stream = C.c_void_p(th.cuda.current_stream().cuda_stream)
        obs2d, act2d, shared, id2, W, ln_w, ln_b, w2, b2, w3, b3 = ctx.saved_tensors
        n = ctx.n_agents
        no, na_ = obs2d.shape[1], act2d.shape[1]
        rows = shared.shape[0] * n
        dev = shared.device
        dq = dq.contiguous()
        dz1 = th.empty(rows, 64, dtype=th.float32, device=dev)                 # (per head, consumed inside its call)
        grads = th.empty(2, 64 * 64 + 64 * 4 + 1, dtype=th.float32, device=dev)
        d_shared = th.empty((2,) + tuple(shared.shape), dtype=th.float32, device=dev)
        d_id = th.empty(2, n, 64, dtype=th.float32, device=dev)
        ws = _critic_workspace(dev)
        for h in range(2):
            g = grads[h]
            args = _critic_args(shared, ln_w, ln_b, w2, b2, w3, b3, ctx.eps)
            args.rows, args.z1, args.z_shared, args.n_agents = rows, None, shared.data_ptr(), n
            _set_critic_ids(args, W, no, n, dense=id2 if h else None)
            args.dq, args.dz1 = dq[h * rows:].data_ptr(), dz1.data_ptr()
            args.d_fc2_w, args.d_fc2_b, args.d_fc3_w = g.data_ptr(), g[4096:].data_ptr(), g[4160:].data_ptr()
            args.d_fc3_b = g[4352:].data_ptr()
            if ln_w is not None:
                args.d_ln_w, args.d_ln_b = g[4224:].data_ptr(), g[4288:].data_ptr()
            args.workspace, args.workspace_floats, args.overwrite_grads = ws.data_ptr(), ws.numel(), 1
            args.d_z_shared, args.d_z_id = d_shared[h].data_ptr(), d_id[h].data_ptr()
            _lib.check(lib.flexnet_critic_tail_backward(C.byref(args), stream), "flexnet_critic_tail_backward")
        gsum = grads[0] + grads[1]
        dsum = d_shared[0] + d_shared[1]
        isum = d_id[0] + d_id[1]
        dW = th.empty_like(W)
        d_bias = th.empty(64, dtype=th.float32, device=dev)
        tall_wgrad(dsum, obs2d, out=dW[:, :no], colsum=d_bias, x2=act2d, out2=dW[:, no + n:no + n + na_])
        dW[:, no:no + n] = isum.t()
        tot = d_id[1, 0]                       # the flag column sees every agent's row of the second head
        for i in range(1, n):
            tot = tot + d_id[1, i]             # (n - 1 pointwise adds: no ATen reduction in a captured graph)
        dW[:, -1] = tot
        if W.shape[1] > no + n + na_ + 1:
            dW[:, no + n + na_:-1] = 0.0
        d_w2, d_b2, d_w3 = gsum[:4096].view(64, 64), gsum[4096:4160], gsum[4160:4224].view(1, 64)
        d_g, d_b, d_b3 = gsum[4224:4288], gsum[4288:4352], gsum[4352:4353]
        has_ln = ln_w is not None
        return (None, None, None, dW, d_bias, (d_g if has_ln else None), (d_b if has_ln else None), d_w2, d_b2, d_w3, d_b3, None)


# Measured and NOT adopted (round 5; VERDICT r04 item 1b asked for the small launches to be folded into their neighbours): the
# value sub-update's small launches BESIDE its matrix work on a second stream — the reward-statistics pass (6.7 us, needs only
# the batch's rewards) while the first-layer product runs, the finish launch (5.4 us: parameter gradients from the partial
# rows, loss, running statistics) while the first layer's weight gradient runs; neither needs the other's output.  Under
# HIP-graph capture the fork and join become graph edges, and every cross-branch edge of a replayed graph costs 4-9 us of idle
# time on this runtime: kernel time 396 -> 370 us per value sub-update, span 404 -> 401 (profiles/r05g4_update_timeline.txt).
# One stream is the default; FLEX_TD_FORK=1 keeps the two-stream form reachable (same kernels, same bits:
# tests/test_critic_gpu.py).
TD_FORK = os.environ.get("FLEX_TD_FORK", "0") == "1"
_SIDE_STREAMS = {}


def _side_stream(device):
    s = _SIDE_STREAMS.get(device)
    if s is None:
        s = _SIDE_STREAMS[device] = th.cuda.Stream(device=device)
    return s


class _CriticTdLossFn(th.autograd.Function):
    """mean((BatchNorm(reward) + gamma (1 - done) Q'(s', pi'(s')) - Q(s, a))^2) for the shared-parameter critic on replayed
    inputs — maddpg.py:100-123 over maddpg.py:33-76 + mlp_critic.py:25-33 — with the critic's backward run IN the forward
    (include/flexnet.h: flexnet_critic_td_backward): the matrix-core backward kernel recomputes the tail's forward anyway,
    so it forms q, the TD error, dLoss/dq and the loss itself.  No forward launch of the tail, no q / dq tensors, no
    td_apply launch; every parameter gradient of the critic is ready when the loss is, and ``backward`` hands them over
    (times the incoming gradient unless that is util.unit_seed).  The BatchNorm module's running statistics move as its
    own training-mode forward would move them."""

    @staticmethod
    def forward(ctx, obs2d, act2d, n_agents, W, bias, ln_w, ln_b, w2, b2, w3, b3, eps, next_q, reward, done, gamma, bn):
        import ctypes as C
        from . import _lib
        lib = _lib.load()
        stream = C.c_void_p(th.cuda.current_stream().cuda_stream)
        n = n_agents
        no, na_ = obs2d.shape[1], act2d.shape[1]
        dev = obs2d.device
        nq, r, d = next_q.reshape(-1, n).contiguous(), reward.contiguous(), done.reshape(-1).contiguous()
        t = _td_args(r, d, nq, gamma, bn, update_stats=True)
        _td_sync_stats(t, _TD_WS[r.device], bn)
        offer = TD_OFFERS.get(r.data_ptr())
        if offer is not None and offer["ref"]() is None:
            del TD_OFFERS[r.data_ptr()]
            offer = None
        if offer is not None and t.normalise and not t.stats_ready:
            t.stats_ready, offer["taken"] = 1, True           # filed by the launch that refreshed this batch (offer_td_stats)
        fork = TD_FORK
        main = th.cuda.current_stream(dev)
        side = _side_stream(dev) if fork else None
        if fork and t.normalise and not t.stats_ready:
            # the statistics pass beside the first-layer product (it needs the rewards only)
            side.wait_stream(main)
            _lib.check(lib.flexnet_td_stats(C.byref(t), C.c_void_p(side.cuda_stream)), "flexnet_td_stats")
            t.stats_ready = 1
        shared = critic_first_layer(bias, obs2d, act2d, W, no + n)
        if fork:
            main.wait_stream(side)
        rows = shared.shape[0] * n
        dz1 = th.empty(rows, 64, dtype=th.float32, device=dev)
        grads = th.empty(64 * 64 + 64 * 4 + 1, dtype=th.float32, device=dev)
        d_w2, d_b2, d_w3 = grads[:4096].view(64, 64), grads[4096:4160], grads[4160:4224].view(1, 64)
        d_g, d_b, d_b3 = grads[4224:4288], grads[4288:4352], grads[4352:4353]
        args = _critic_args(shared, ln_w, ln_b, w2, b2, w3, b3, eps)
        args.rows, args.z1, args.z_shared, args.n_agents = rows, None, shared.data_ptr(), n
        _set_critic_ids(args, W, no, n)
        args.dz1 = dz1.data_ptr()
        args.d_fc2_w, args.d_fc2_b, args.d_fc3_w, args.d_fc3_b = d_w2.data_ptr(), d_b2.data_ptr(), d_w3.data_ptr(), d_b3.data_ptr()
        if ln_w is not None:
            args.d_ln_w, args.d_ln_b = d_g.data_ptr(), d_b.data_ptr()
        ws = _critic_workspace(dev)
        args.workspace, args.workspace_floats, args.overwrite_grads = ws.data_ptr(), ws.numel(), 1
        d_shared = th.empty_like(shared)
        dW = th.empty_like(W)
        args.d_z_shared, args.d_z_id = d_shared.data_ptr(), dW[:, no:no + n].data_ptr()
        args.d_z_id_agent_stride, args.d_z_id_unit_stride = 1, W.shape[1]
        loss = th.empty((), dtype=th.float32, device=dev)
        t.loss = loss.data_ptr()
        d_bias = th.empty(64, dtype=th.float32, device=dev)
        sm = args.variant_pgrad32 == 0          # (the 16-row kernel writes d_z_shared itself: the finish does not feed the weight gradient)
        if fork and sm:
            _lib.check(lib.flexnet_critic_td_backward_phases(C.byref(args), C.byref(t), 1, stream), "flexnet_critic_td_backward")
            side.wait_stream(main)
            _lib.check(lib.flexnet_critic_td_backward_phases(C.byref(args), C.byref(t), 2, C.c_void_p(side.cuda_stream)),
                       "flexnet_critic_td_backward")
            tall_wgrad(d_shared, obs2d, out=dW[:, :no], colsum=d_bias, x2=act2d, out2=dW[:, no + n:no + n + na_])
            main.wait_stream(side)
        elif sm:
            # the finish (phase 2) depends on the backward kernel alone: it rides in the weight gradient's second-stage launch
            _lib.check(lib.flexnet_critic_td_backward_phases(C.byref(args), C.byref(t), 1, stream), "flexnet_critic_td_backward")
            tall_wgrad(d_shared, obs2d, out=dW[:, :no], colsum=d_bias, x2=act2d, out2=dW[:, no + n:no + n + na_],
                       critic_finish=(args, t))
        else:
            _lib.check(lib.flexnet_critic_td_backward(C.byref(args), C.byref(t), stream), "flexnet_critic_td_backward")
            tall_wgrad(d_shared, obs2d, out=dW[:, :no], colsum=d_bias, x2=act2d, out2=dW[:, no + n:no + n + na_])
        if W.shape[1] > no + n + na_:
            dW[:, no + n + na_:] = 0.0
        has_ln = ln_w is not None
        ctx.has_ln = has_ln
        ctx.save_for_backward(dW, d_bias, grads)
        return loss

    @staticmethod
    def backward(ctx, g):
        from .util import is_unit_seed
        dW, d_bias, grads = ctx.saved_tensors
        if not is_unit_seed(g):
            dW, d_bias, grads = dW * g, d_bias * g, grads * g
        d_w2, d_b2, d_w3 = grads[:4096].view(64, 64), grads[4096:4160], grads[4160:4224].view(1, 64)
        d_g, d_b, d_b3 = grads[4224:4288], grads[4288:4352], grads[4352:4353]
        return (None, None, None, dW, d_bias, (d_g if ctx.has_ln else None), (d_b if ctx.has_ln else None), d_w2, d_b2, d_w3,
                d_b3) + (None,) * 6


def _td_args(reward, done, next_q, gamma, bn, update_stats=True):
    """FlexTdLossArgs for [rows, n] tensors (q / dq / loss left to the caller); the workspace is per device."""
    from . import _lib
    rows, n = reward.shape
    if reward.device not in _TD_WS:
        _TD_WS[reward.device] = th.empty(_lib.FLEXNET_TD_WS_FLOATS // 2, dtype=th.float64, device=reward.device)
    ws = _TD_WS[reward.device]
    a = _lib.FlexTdLossArgs()
    a.rows, a.n_agents, a.normalise, a.gamma = rows, n, int(bn is not None), float(gamma)
    a.reward, a.done, a.next_q = reward.data_ptr(), done.data_ptr(), next_q.data_ptr()
    if bn is not None:
        a.bn_eps, a.bn_momentum = float(bn.eps), float(bn.momentum)
        if bn.affine:
            a.bn_weight, a.bn_bias = bn.weight.data_ptr(), bn.bias.data_ptr()
        if bn.track_running_stats and update_stats:
            a.running_mean, a.running_var = bn.running_mean.data_ptr(), bn.running_var.data_ptr()
            a.num_batches_tracked = bn.num_batches_tracked.data_ptr()
    a.workspace, a.workspace_floats = ws.data_ptr(), 2 * ws.numel()
    return a


CRITIC_TD_MIN_ROWS = 65536          # csrc/critic.hip CRITIC_MFMA_MIN_ROWS: below it the matrix-core backward is not used


def critic_td_loss_supported(critic, obs2d, act2d, n_agents, next_q, reward, done, bn):
    """What flexnet_critic_td_backward covers: the replayed-input critic node's configuration at matrix-core batch sizes,
    an fc1 without trailing extra columns, the fused value loss's tensors, gradients wanted."""
    W = critic.fc1.weight
    return (critic_replayed_supported(critic, obs2d, act2d, n_agents) and th.is_grad_enabled() and W.requires_grad
            and obs2d.shape[0] * n_agents >= CRITIC_TD_MIN_ROWS and CRITIC_VARIANT == 0
            and W.shape[1] == obs2d.shape[1] + n_agents + act2d.shape[1]
            and reward.dim() == 2 and reward.shape == (obs2d.shape[0], n_agents)
            and td_loss_supported(reward, next_q, reward, done, bn))


def critic_td_loss(obs2d, act2d, n_agents, critic, next_q, reward, done, gamma, bn):
    ln = critic.layernorm if critic.args.layernorm else None
    return _CriticTdLossFn.apply(obs2d, act2d, n_agents, critic.fc1.weight, critic.fc1.bias,
                                 None if ln is None else ln.weight, None if ln is None else ln.bias,
                                 critic.fc2.weight, critic.fc2.bias, critic.fc3.weight, critic.fc3.bias,
                                 1e-5 if ln is None else ln.eps, next_q, reward, done, gamma, bn)


class _ExpandAgentsFn(th.autograd.Function):
    """x [b, 1, a] -> [b, n, a] (the agent-summed action of matd3.py:92-97 / iddpg.py:66-71 handed to every agent); the
    backward sums over the agent axis with n - 1 pointwise adds instead of ATen's reduce_kernel (a captured HIP graph
    must not hold one)."""

    @staticmethod
    def forward(ctx, x, n):
        ctx.n = n
        return x.expand(-1, n, -1)

    @staticmethod
    def backward(ctx, g):
        parts = g.unbind(1)
        tot = parts[0]
        for p in parts[1:]:
            tot = tot + p
        return tot.unsqueeze(1), None


def expand_agents(x, n):
    return _ExpandAgentsFn.apply(x, n) if x.requires_grad else x.expand(-1, n, -1)


class _CriticPolicyFn(th.autograd.Function):
    """The shared-parameter critic inside the POLICY loss, maddpg.py:33-76 + maddpg.py:104-107: q[b, i] on the policy's own
    actions, differentiated w.r.t. those actions only (row i of a sample carries agent i's own action block; the other
    agents' blocks are detached, maddpg.py:47-54).  Value-wise every row is still shared[b] + W_id[:, i] — the own-action
    term is zero-valued — so the forward is two GEMMs and the composed tail kernel; the backward is the tail's dz1-only
    kernel and d act[b, i] = dz1[b, i] @ W_act[:, block i].  The critic's parameters get no gradient from this node: it
    is used where only the policy optimiser steps (utils/trainer.py:99-108 steps the policy on the policy loss)."""

    @staticmethod
    def forward(ctx, obs2d, act, W, bias, ln_w, ln_b, w2, b2, w3, b3, eps):
        import ctypes as C
        from . import _lib
        lib = _lib.load()
        b, n, na_ = act.shape
        no = obs2d.shape[1]
        act2d = act.reshape(b, n * na_)
        shared = critic_first_layer(bias, obs2d, act2d, W, no + n)
        rows = b * n
        q = th.empty(rows, 1, dtype=th.float32, device=shared.device)
        args = _critic_args(shared, ln_w, ln_b, w2, b2, w3, b3, eps)
        args.rows, args.z1, args.z_shared, args.n_agents = rows, None, shared.data_ptr(), n
        _set_critic_ids(args, W, no, n)
        args.q = q.data_ptr()
        _lib.check(lib.flexnet_critic_tail_forward(C.byref(args), C.c_void_p(th.cuda.current_stream().cuda_stream)),
                   "flexnet_critic_tail_forward")
        ctx.eps, ctx.dims = eps, (b, n, na_, no)
        ctx.save_for_backward(shared, W, ln_w, ln_b, w2, b2, w3, b3)
        return q

    @staticmethod
    def backward(ctx, dq):
        import ctypes as C
        from . import _lib
        lib = _lib.load()
        shared, W, ln_w, ln_b, w2, b2, w3, b3 = ctx.saved_tensors
        b, n, na_, no = ctx.dims
        rows = b * n
        dq = dq.contiguous()
        dz1 = th.empty(rows, 64, dtype=th.float32, device=shared.device)
        args = _critic_args(shared, ln_w, ln_b, w2, b2, w3, b3, ctx.eps)
        args.rows, args.z1, args.z_shared, args.n_agents = rows, None, shared.data_ptr(), n
        _set_critic_ids(args, W, no, n)
        args.dq, args.dz1 = dq.data_ptr(), dz1.data_ptr()
        _lib.check(lib.flexnet_critic_tail_backward(C.byref(args), C.c_void_p(th.cuda.current_stream().cuda_stream)),
                   "flexnet_critic_tail_backward")
        W_act = W[:, no + n:no + n + n * na_].reshape(64, n, na_)
        d_act = th.einsum("bih,hia->bia", dz1.view(b, n, 64), W_act)
        return (None, d_act) + (None,) * 9


class _CriticPolicyLossFn(th.autograd.Function):
    """sign * mean(Q(s, pi(s))) over all [b, n] entries for the shared-parameter critic, differentiated w.r.t. the policy's
    actions only — the policy loss of maddpg.py:104-107 as ONE node.  The gradient of a mean is a constant, so the critic
    needs no forward pass of its own: the dz1-only backward kernel (which recomputes the forward) runs in this node's
    ``forward`` with a uniform dLoss/dq and also returns the sum of q, i.e. the loss.  No tail forward launch, no loss
    reduction launches; ``backward`` hands the stored action gradient over."""

    @staticmethod
    def forward(ctx, obs2d, act, W, bias, ln_w, ln_b, w2, b2, w3, b3, eps, sign):
        import ctypes as C
        from . import _lib
        lib = _lib.load()
        b, n, na_ = act.shape
        no = obs2d.shape[1]
        act2d = act.reshape(b, n * na_)
        shared = critic_first_layer(bias, obs2d, act2d, W, no + n)
        rows = b * n
        dz1 = th.empty(rows, 64, dtype=th.float32, device=shared.device)
        loss = th.empty((), dtype=th.float32, device=shared.device)
        args = _critic_args(shared, ln_w, ln_b, w2, b2, w3, b3, eps)
        args.rows, args.z1, args.z_shared, args.n_agents = rows, None, shared.data_ptr(), n
        _set_critic_ids(args, W, no, n)
        args.dz1 = dz1.data_ptr()
        args.dq_uniform, args.dq_value, args.q_mean_scale, args.q_mean_out = 1, sign / rows, sign / rows, loss.data_ptr()
        ws = _critic_workspace(shared.device)
        args.workspace, args.workspace_floats = ws.data_ptr(), ws.numel()
        _lib.check(lib.flexnet_critic_tail_backward(C.byref(args), C.c_void_p(th.cuda.current_stream().cuda_stream)),
                   "flexnet_critic_tail_backward")
        W_act = W[:, no + n:no + n + n * na_].reshape(64, n, na_)
        d_act = th.einsum("bih,hia->bia", dz1.view(b, n, 64), W_act)
        ctx.save_for_backward(d_act)
        return loss

    @staticmethod
    def backward(ctx, g):
        from .util import is_unit_seed
        (d_act,) = ctx.saved_tensors
        return (None, d_act if is_unit_seed(g) else d_act * g) + (None,) * 10


def critic_policy_loss_supported(critic, obs2d, act, n_agents):
    return (critic_policy_supported(critic, obs2d, act, n_agents) and CRITIC_VARIANT == 0
            and act.shape[0] * n_agents >= CRITIC_TD_MIN_ROWS and act.requires_grad and th.is_grad_enabled())


def critic_policy_loss(obs2d, act, critic, sign=-1.0):
    ln = critic.layernorm if critic.args.layernorm else None
    return _CriticPolicyLossFn.apply(obs2d, act, critic.fc1.weight, critic.fc1.bias,
                                     None if ln is None else ln.weight, None if ln is None else ln.bias,
                                     critic.fc2.weight, critic.fc2.bias, critic.fc3.weight, critic.fc3.bias,
                                     1e-5 if ln is None else ln.eps, float(sign))


def critic_policy_supported(critic, obs2d, act, n_agents):
    a = critic.args
    W = critic.fc1.weight
    return (obs2d.is_cuda and obs2d.dtype == th.float32 and act.dtype == th.float32 and act.dim() == 3 and a.hid_size == 64
            and a.hid_activation == "relu" and critic.fc3.out_features == 1 and getattr(critic, "fused_tail", True)
            and 1 <= n_agents <= 8 and not obs2d.requires_grad and act.shape[1] == n_agents
            and W.shape[1] in (obs2d.shape[1] + n_agents + n_agents * act.shape[2],
                               obs2d.shape[1] + n_agents + n_agents * act.shape[2] + 1))      # + MATD3's twin flag column


def critic_replayed_supported(critic, obs2d, act2d, n_agents):
    a = critic.args
    W = critic.fc1.weight
    return (obs2d.is_cuda and obs2d.dtype == th.float32 and act2d.dtype == th.float32 and a.hid_size == 64
            and a.hid_activation == "relu" and critic.fc3.out_features == 1 and getattr(critic, "fused_tail", True)
            and 1 <= n_agents <= 8 and not obs2d.requires_grad and not act2d.requires_grad
            and obs2d.shape[0] >= WGRAD_MIN_ROWS and W.shape[1] in (obs2d.shape[1] + n_agents + act2d.shape[1],
                                                                    obs2d.shape[1] + n_agents + act2d.shape[1] + 1)
            and all(x.dim() == 2 and x.stride(1) == 1 and x.shape[1] <= x.stride(0) < (1 << 24) for x in (obs2d, act2d)))


class CriticTail:
    @staticmethod
    def apply_policy(obs2d, act, critic):
        ln = critic.layernorm if critic.args.layernorm else None
        return _CriticPolicyFn.apply(obs2d, act, critic.fc1.weight, critic.fc1.bias,
                                     None if ln is None else ln.weight, None if ln is None else ln.bias,
                                     critic.fc2.weight, critic.fc2.bias, critic.fc3.weight, critic.fc3.bias,
                                     1e-5 if ln is None else ln.eps)

    @staticmethod
    def apply_replayed_twin(obs2d, act2d, n_agents, critic):
        ln = critic.layernorm if critic.args.layernorm else None
        return _CriticReplayedTwinFn.apply(obs2d, act2d, n_agents, critic.fc1.weight, critic.fc1.bias,
                                           None if ln is None else ln.weight, None if ln is None else ln.bias,
                                           critic.fc2.weight, critic.fc2.bias, critic.fc3.weight, critic.fc3.bias,
                                           1e-5 if ln is None else ln.eps)

    @staticmethod
    def apply_replayed(obs2d, act2d, n_agents, critic, twin=False):
        ln = critic.layernorm if critic.args.layernorm else None
        return _CriticReplayedFn.apply(obs2d, act2d, n_agents, critic.fc1.weight, critic.fc1.bias,
                                       None if ln is None else ln.weight, None if ln is None else ln.bias,
                                       critic.fc2.weight, critic.fc2.bias, critic.fc3.weight, critic.fc3.bias,
                                       1e-5 if ln is None else ln.eps, twin)

    @staticmethod
    def apply_composed(shared, id_cols, critic):
        ln = critic.layernorm if critic.args.layernorm else None
        return _CriticTailComposedFn.apply(shared, id_cols, None if ln is None else ln.weight,
                                           None if ln is None else ln.bias, critic.fc2.weight, critic.fc2.bias,
                                           critic.fc3.weight, critic.fc3.bias, 1e-5 if ln is None else ln.eps)

    @staticmethod
    def apply(z1, critic):
        ln = critic.layernorm if critic.args.layernorm else None
        return _CriticTailFn.apply(z1, None if ln is None else ln.weight, None if ln is None else ln.bias,
                                   critic.fc2.weight, critic.fc2.bias, critic.fc3.weight, critic.fc3.bias,
                                   1e-5 if ln is None else ln.eps)
