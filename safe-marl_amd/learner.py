"""MADDPG / SAFEMADDPG on device-resident batches, with the reference's class and method names
(madrl/models/model.py, maddpg.py, safemaddpg.py) so that ``PGTrainer(args, MADDPG, env, logger)``
and reference state_dicts keep working.

What is different from the reference, by design:
  * every method takes batches of any size (thousands of envs), never assumes batch 1;
  * a replay batch is already a tuple of device tensors (replay_buffer.DeviceReplayBuffer): there is no
    numpy -> tensor -> device ``unpack_data`` round trip per sub-update (model.py:308-323);
  * the centralised critic does not materialise its [B*n, n*(o+a)+n] input (maddpg.py:33-76); it forms
    fc1's output from the column blocks of fc1.weight — the observation block once per sample instead
    of n times — which is the same affine map up to fp32 summation order;
  * the rollout steps N environments per iteration without a host sync (``train_process`` on a
    VecFlexProvisionEnv) and the soft target update is one fused ``torch._foreach`` call.
The N=1 path through ``FlexibilityProvisionEnv`` reproduces the reference's cadence exactly.
"""
from __future__ import annotations

import os

import numpy as np
import torch as th
import torch.nn as nn

from .nets import (WGRAD_MIN_ROWS, CriticTail, critic_first_layer, critic_td_loss, critic_td_loss_supported, critic_policy_loss,
                   critic_policy_loss_supported, expand_agents, MLPAgent, MLPCritic, RNNAgent, critic_policy_supported, critic_replayed_supported,
                   critic_tail_supported, fused_actor_forward, tall_linear, td_loss, td_loss_supported, wide_batch_linear,
                   batchnorm_stats_supported, batchnorm_update_running_stats, sync_batchnorm)
from .replay_buffer import Transition
from .util import graph_capture, prep_obs, scale_action, select_action, translate_action, mean_all


class RolloutGraph:
    """One vector step of the rollout — policy forward, exploration noise, action scaling, the fused env kernel,
    the transition into the slab replay ring, statistics, hand-over of the hidden state — captured ONCE as a HIP graph
    and replayed per step.  Eagerly this is ~25 tiny kernels whose launch cost (0.36 ms) dwarfs the 14 us env kernel; as
    a graph it is one launch.  Everything the graph touches is a static tensor owned here, by the env or by the ring.

    The observation the policy reads IS the env's own output buffer (``env.obs``): the ring already holds it (the pack
    kernel of the step that produced it wrote it into the next slab), so nothing is handed over or copied twice."""

    def __init__(self, model, env, buf):
        self.model, self.env, self.buf = model, env, buf
        N, n, o, a, h = env.n_envs, model.n_, model.obs_dim, model.act_dim, model.hid_dim
        dev = model.device
        # Row mode (round 4): where the env's step kernel files observations itself (ring I/O below) the replay keeps every
        # feature ROW once and the policy kernels read the stacked observation in place from the env's own history —
        # nothing copies a [n, 6 H] window per step any more; elsewhere (general bodies) slabs hold stacked observations
        self.history = getattr(getattr(env, "vec", env), "history", None)
        self.rows_capable = bool(hasattr(env, "obs_source") and hasattr(env, "set_obs_ring") and env.obs.is_cuda
                                 and self.history and o == 6 * self.history)
        if not buf.slab_mode:
            if buf.store is not None:
                raise RuntimeError("replay buffer already holds field-by-field transitions; the graph rollout needs slab mode")
            buf.alloc_slabs(N, n, o, a, h)                    # (re-made in row mode by _configure_env where that applies)
        if (buf.n_envs, buf.n_agents, buf.obs_dim, buf.act_dim, buf.hid_dim) != (N, n, o, a, h):
            raise RuntimeError("replay buffer was allocated for another environment batch")
        self._obs = env.obs                         # [N, n, o]: written by the env kernel, read by the policy (tensor mode)
        self._hid = th.zeros(N, n, h, device=dev)
        self._info_sum = th.zeros(env.info.shape[1], dtype=th.float64, device=dev)
        self._rew_sum = th.zeros((), dtype=th.float64, device=dev)
        self._fail_sum = th.zeros((), dtype=th.float64, device=dev)
        self.std = float(model.args.fixed_policy_std)
        self.graph = None
        self.bursts = {}                                       # k -> graph of k bodies (run())
        self._torch_noise = False
        self.rng_state = th.zeros(2, dtype=th.int64, device=dev)           # [seed, step] of the actor kernel's noise stream
        self.rng_state[0] = int(th.randint(0, 2 ** 62, (1,)).item())
        self.plain = type(model).get_actions is MADDPG.get_actions and bool(model.args.action_enforcebound)
        self.avail = th.ones(N, n, a, device=dev)             # every action is available (env:721-730)
        # MATD3 / IDDPG with the bound enforced: their agent-summed action selection (matd3.py:92-97, iddpg.py:66-71 over
        # util.py:57-64) and translate_action as ONE launch behind the fused policy — bit-identical to get_actions +
        # env_action (same draws from torch's generator, every fp32 rounding in the same place), ~40 launches fewer per step
        self.summed = (type(model).__name__ in ("MATD3", "IDDPG")
                       and type(model).get_actions in (MATD3.get_actions, IDDPG.get_actions)
                       and bool(model.args.action_enforcebound) and bool(model.args.continuous) and a > 1
                       and env.obs.is_cuda and model.fused_inference and model.args.shared_params)
        if self.summed:
            with th.no_grad():                  # exp(sum over agents of the fixed log-std): the tensor ops' own arithmetic
                ls = model._log_stds_like(th.zeros(1, n, a, device=dev))
                self.std_sum = _sum_agents(ls.expand(1, n, a)).exp().reshape(a).to(th.float32).contiguous()
            self.act_pol_buf = th.zeros(N, n, a, device=dev)
            self.env_act_buf = th.zeros(N, n, a, device=dev)
        # plain MADDPG on the GPU: policy + exploration in one HIP launch, ring write + hand-over + statistics in another
        self.safe = type(model).__name__ == "SAFEMADDPG"       # + the safety projection between policy and env
        # (the actor kernel's exploration epilogue IS tanh(mean + std * noise), util.py:57-64: without action_enforcebound
        # the reference adds unbounded noise, util.py:66-74, and the general body below runs select_action itself)
        self._fast = (type(model).__name__ in ("MADDPG", "SAFEMADDPG") and model.fused_inference
                     and model.args.shared_params and env.obs.is_cuda and model.args.agent_type == "rnn" and h == 64
                     and o <= 144 and bool(model.args.action_enforcebound))
        # ring write / hand-over / statistics in ONE launch of this project's kernel (fixed-order block sums): no ATen
        # reduction is ever captured into the rollout graph, whatever the algorithm
        self.packable = env.obs.is_cuda and h == 64 and o <= 144 and n <= 8 and a <= 8 and (n * o) % 4 == 0
        if self.safe:
            self.predictor = tuple(th.as_tensor(x, dtype=th.float64, device=dev).contiguous() for x in model.predictor)
        self.env_calls = None                                  # env.calls after this object's last step (continuity check)
        # The env's step kernel advances the slab the step fills (cursor[1]) itself (one lane): no launch of its own.
        self.cursor_stepped = 1 if (hasattr(env, "set_step_counter") and env.obs.is_cuda) else 0
        # Ring I/O (the fused path): the env kernel writes its observation straight into the slab after the cursor, the
        # actor kernel reads observation and hidden state from the slab at the cursor — the observation is written once,
        # where the replay keeps it, and never copied.
        self.ring_io = bool(self._fast and self.cursor_stepped and self.rows_capable)
        # Sink (ring I/O + an env that files transitions): the env step kernel also writes the small record, the masked
        # hidden state and the per-environment statistics — a vector step is TWO launches, policy and environment, and no
        # bookkeeping kernel.  Cursor cells then follow the two-kernel protocol of include/flexenv.h: cursor[0] is read by
        # the policy and written by the env step (slab to read next), cursor[1] is read by the env step and written by the
        # policy (slab it just read).
        # ... within the sink's own limits (flexenv_set_replay_sink: action row <= 32 floats, recurrent row <= 384 floats —
        # seven agents x 64 units is 448 — and the two-environments-per-wavefront instantiation, i.e. <= 32 PQ buses);
        # outside them the pack kernel files the transition (three launches per step)
        n_bus = getattr(getattr(env, "vec", env), "n_bus", None)       # (a wrapped env without it: not eligible — pack path)
        two_per_wave = n_bus is not None and n_bus - 1 <= 32
        self.sink = bool(self.ring_io and hasattr(env, "set_replay_sink") and n * a <= 32 and n * h <= 384 and two_per_wave)
        # MATD3 / IDDPG with the one-launch action selection (`summed`): the same ring I/O and sink — policy kernel (slab at the
        # cursor -> means, new hidden state), agent_sum_explore_kernel (-> the action the replay keeps, the env's action), env
        # step (files the transition): three launches, no pack kernel, the observation written once (round 3)
        self.summed_sink = bool(self.summed and self.cursor_stepped and self.rows_capable
                                and hasattr(env, "set_replay_sink") and n * a <= 32 and n * h <= 384
                                and two_per_wave and h == 64 and o <= 144
                                and os.environ.get("FLEX_SUMMED_SINK", "1") != "0")
        if self.sink or self.summed_sink:
            self.act_buf = th.zeros(N * n, a, device=dev)
            self.hid_buf = th.zeros(N * n, h, device=dev)
            self.acc = th.zeros(N, 10, dtype=th.float64, device=dev)
        # Burst launch (round 3): with the sink, plain MADDPG (nothing between policy and environment) and at most five
        # agents — a block's sixteen environments are then at most five 16-row policy tiles, what one CU's LDS-resident
        # weights serve — run(m) is ONE persistent launch per m steps (include/flexenv.h: flexenv_rollout_burst).
        # SAFEMADDPG: the safety projection is a phase of the same launch (each wavefront projects its own two environments)
        self.burst_launch = bool(self.sink and n <= 5 and o % 4 == 0 and hasattr(getattr(env, "vec", env), "rollout_burst")
                                 and (not self.safe or a == 4) and os.environ.get("FLEX_ROLLOUT_BURST", "1") != "0")
        if self.burst_launch:
            self.means_buf = th.zeros(N * n, a, device=dev)
            self.burst_env_act = th.zeros(N * n, a, device=dev)
            if self.safe:
                self.burst_adjusted = th.zeros(N, 4 * n, dtype=th.float64, device=dev)
                self.burst_safe_env_act = th.zeros(N, 4 * n, device=dev)
        self._configure_env()

    def _configure_env(self):
        """Point the env's device-side hooks at this object's ring for the mode in force (fast / general body, in-kernel
        or torch noise): called at construction and whenever a test flips ``fast`` or ``torch_noise``."""
        env, buf = self.env, self.buf
        if not self.cursor_stepped:
            return
        N, n, o = env.n_envs, self.model.n_, self.model.obs_dim
        # the replay's layout follows the mode: feature rows where the env files observations itself, stacked slabs otherwise
        if bool(self.ring_active) != bool(buf.row_mode):
            if buf.k != 0 or buf.length != 0:
                raise RuntimeError("the rollout body changed between ring I/O and tensor hand-over with transitions in the replay")
            buf.alloc_slabs(N, n, o, self.model.act_dim, self.model.hid_dim, history=self.history if self.ring_active else None)
        if self.ring_active:
            self.obs_src = env.obs_source()                 # the policy kernels read the stacked observation in place
        row_stride = N * n * buf.ROW_W
        if self.sink_active:
            env.set_step_counter(None)                      # the env step sets cursor[0] itself in this mode
            env.set_obs_ring(buf.cursor[1:], row_stride, buf.slabs)
            env.set_replay_sink(self.act_buf, self.hid_buf, buf.small_ring, buf.hid_ring, self.acc, cursor_out=buf.cursor[0:1],
                                aux_counter=None if (self._torch_noise or self.summed_sink) else self.rng_state[1:2])
        else:
            env.set_step_counter(buf.cursor[1:], buf.slabs)
            if self.ring_active:
                env.set_obs_ring(buf.cursor, row_stride, buf.slabs)
            if hasattr(env, "set_replay_sink"):
                env.set_replay_sink(None, None, None, None, None)

    @property
    def fast(self):
        return self._fast

    @fast.setter
    def fast(self, value):
        self._fast = bool(value)
        self._configure_env()

    @property
    def torch_noise(self):
        return self._torch_noise

    @torch_noise.setter
    def torch_noise(self, value):
        self._torch_noise = bool(value)
        self._configure_env()

    @property
    def sink_active(self):
        return (self.sink and self.fast) or self.summed_sink

    @property
    def fused_burst(self):
        return (self.burst_launch and self.sink_active and not self._torch_noise
                and not (self.safe and self.model.intended_actions))       # (that routing transposes in torch)

    # episode statistics: block sums of the pack kernel, or (sink) the env step's per-environment running sums, added up
    # when asked for (outside any graph)
    @property
    def info_sum(self):
        return self.acc[:, :self._info_sum.numel()].sum(0) if self.sink_active else self._info_sum

    @property
    def rew_sum(self):
        return self.acc[:, 7].sum() if self.sink_active else self._rew_sum

    @property
    def fail_sum(self):
        return self.acc[:, 8].sum() if self.sink_active else self._fail_sum

    @property
    def ring_active(self):
        # (tests switch `fast` off to run the general body on the same object)
        return (self.ring_io and self.fast) or self.summed_sink

    @property
    def obs(self):
        """The observation the next policy evaluation reads: [N, n, o]."""
        if self.ring_active:                                # (a materialised copy: the policy kernels read the env's history)
            return self.env.obs_view().view(self.env.n_envs, self.model.n_, self.model.obs_dim)
        return self._obs

    @property
    def hid(self):
        """The hidden state the next policy evaluation starts from: [N, n, h]."""
        if self.ring_active:
            return self.buf.hid_ring[self.buf.k % self.buf.slabs].view(self.env.n_envs, self.model.n_, self.model.hid_dim)
        return self._hid

    def _pack(self, action, hid):
        """model.py:230-262 for every environment in one launch (include/flexnet.h: flexnet_rollout_pack)."""
        import ctypes as C
        from . import _lib
        m, env, buf = self.model, self.env, self.buf
        a = _lib.FlexRolloutPackArgs()
        a.n_envs, a.n_agents, a.obs_dim, a.act_dim = env.n_envs, m.n_, m.obs_dim, m.act_dim
        a.slabs, a.small_w, a.info_w, a.cursor_stepped = buf.slabs, buf.small_w, env.info.shape[1], self.cursor_stepped
        for name, t in (("action", action), ("reward", env.reward), ("obs_next", None if self.ring_active else env.obs),
                        ("done", env.done), ("hid_new", hid), ("info", env.info), ("failed", env.failed),
                        ("obs_ring", buf.row_ring if buf.row_mode else buf.obs_ring), ("hid_ring", buf.hid_ring),
                        ("small_ring", buf.small_ring),
                        ("hid_state", None if self.ring_active else self._hid), ("cursor", buf.cursor),
                        ("info_sum", self._info_sum), ("rew_sum", self._rew_sum), ("fail_sum", self._fail_sum)):
            if t is not None:
                assert t.is_contiguous()
                setattr(a, name, t.data_ptr())
        if not self.torch_noise:
            a.rng_state = self.rng_state.data_ptr()       # next step of the actor kernel's noise stream
        _lib.check(_lib.load().flexnet_rollout_pack(C.byref(a), C.c_void_p(th.cuda.current_stream().cuda_stream)),
                   "flexnet_rollout_pack")

    def last_transition(self):
        """Views of the transition the last step wrote (tests, debugging): the slab before the cursor."""
        buf = self.buf
        N = self.env.n_envs
        return buf.slab_window((buf.k - 1) * N, N)

    def body(self, burst=0):
        """One vector step (``burst`` = 0), or ``burst`` of them in the fused launch (``fused_burst`` configurations only)."""
        m, env = self.model, self.env
        N = env.n_envs
        if burst and not self.fused_burst:
            raise RuntimeError("rollout burst launch outside its configuration")
        if self.fast:
            with th.no_grad():
                # exploration noise: drawn by the actor kernel from its own Philox stream (seeded from torch's generator
                # when this object was built; the pack kernel advances the step counter), or — torch_noise, used by the
                # tests that compare with the PyTorch glue draw for draw — by torch.randn.  Inside a HIP graph torch.randn
                # costs a launch plus two seed/offset fill kernels per replay.
                buf = self.buf
                noise = th.randn(N, m.n_, m.act_dim, device=env.obs.device) if self.torch_noise else None
                # ring I/O: the hidden state from the slab at the cursor, the stacked observation IN PLACE from the env's history
                ring = dict(ring_cursor=buf.cursor, obs_slab_stride=0, hid_slab_stride=buf.hid_ring.stride(0),
                            obs_source=self.obs_src) if self.ring_active else {}
                obs_in = self._obs                           # (ring I/O: only its shape is used)
                hid_in = buf.hid_ring[0].view(N, m.n_, m.hid_dim) if self.ring_active else self._hid
                if self.sink_active:           # static outputs the env step reads back (registered with its replay sink)
                    ring.update(cursor_out=buf.cursor[1:], out=dict(hidden_out=self.hid_buf, action=self.act_buf))
                if burst:
                    # policy and environment for `burst` steps in one persistent launch (include/flexenv.h:
                    # flexenv_rollout_burst): the policy call's arguments, handed to the env's launch instead of its own
                    ring["out"].update(means=self.means_buf, env_action=self.burst_env_act)
                    vec = env.vec if hasattr(env, "vec") else env
                    safety = None
                    if self.safe:          # safemaddpg.py:90-111 inside the launch: what the three-launch body below calls
                        s_p, s_q, beta = self.predictor
                        safety = dict(s_p=s_p, s_q=s_q, beta=beta, v_min=m.V_min, v_max=m.V_max, adjusted=self.burst_adjusted,
                                      env_action=self.burst_safe_env_act, act_low=m.args.action_low, act_high=m.args.action_high)
                    ring.update(ring_slabs=buf.slabs,
                                launch=lambda args: vec.rollout_burst(args, burst, buf.row_ring, safety=safety))
                out = fused_actor_forward(m.policy_dicts[0], obs_in, hid_in, m.n_, m.args.agent_id, noise=noise,
                                          std=self.std, low=m.args.action_low, high=m.args.action_high,
                                          rng_state=None if self.torch_noise else self.rng_state, **ring)
                if out is None:
                    raise RuntimeError("the fused actor kernel declined a configuration RolloutGraph.fast admitted")
                if burst:
                    return
                _, hid, action, env_action = out
                if self.safe:
                    # safemaddpg.py:90-111: the proposed action goes through the safety layer (HIP closed form,
                    # flexenv_safety_project); the replay keeps the policy's own action (model.py:232)
                    vec = env.vec if hasattr(env, "vec") else env
                    # ... and the env's action (translate_action of the adjusted vector) comes out of the same launch
                    if m.intended_actions:        # (not the reference's routing: SAFEMADDPG.intended_actions)
                        adj, _ = vec.safety_project(action.view(N, m.n_, m.act_dim), *self.predictor, m.V_min, m.V_max,
                                                    want_hit=False)
                        env_action = m.env_action(adj)
                    else:
                        _, _, env_action = vec.safety_project(action.view(N, m.n_, m.act_dim), *self.predictor, m.V_min, m.V_max,
                                                              env_action_range=(m.args.action_low, m.args.action_high),
                                                              want_hit=False)
                env.step(env_action.view(N, m.n_, m.act_dim), fuse_obs=True, auto_reset=True,
                         obs_ring=buf.row_ring if self.ring_active else None, replay_sink=self.sink_active)
                if not self.sink_active:
                    self._pack(action, hid)
                return
        if self.summed and self.summed_sink:
            with th.no_grad():
                buf = self.buf
                out = fused_actor_forward(m.policy_dicts[0], self._obs,
                                          buf.hid_ring[0].view(N, m.n_, m.hid_dim), m.n_, m.args.agent_id,
                                          ring_cursor=buf.cursor, obs_slab_stride=0, obs_source=self.obs_src,
                                          hid_slab_stride=buf.hid_ring.stride(0), cursor_out=buf.cursor[1:],
                                          out=dict(hidden_out=self.hid_buf))
                if out is None:
                    raise RuntimeError("the fused actor kernel declined a configuration RolloutGraph.summed_sink admitted")
                summed_exploration(m, out[0].view(N, m.n_, m.act_dim), env_action=self.env_act_buf,
                                   action_out=self.act_buf.view(N, m.n_, m.act_dim))
                env.step(self.env_act_buf, fuse_obs=True, auto_reset=True, obs_ring=buf.row_ring, replay_sink=True)
            return
        if self.summed:
            with th.no_grad():
                means, _, hid = m.policy(self.obs, last_hid=self.hid)
                summed_exploration(m, means.to(th.float32), env_action=self.env_act_buf, action_out=self.act_pol_buf)
                env.step(self.env_act_buf, fuse_obs=True, auto_reset=True)
                self._pack(self.act_pol_buf, hid.reshape(N, m.n_, -1).to(th.float32).contiguous())
            return
        with th.no_grad():
            if self.plain:
                means, _, hid = m.policy(self.obs, last_hid=self.hid)
                action = action_pol = th.tanh(means + self.std * th.randn_like(means))          # util.py:57-64
            else:
                # MATD3 / IDDPG (agent-summed action selection, matd3.py:88-111, iddpg.py:66-71): their own get_actions,
                # exactly as the eager loop calls it; the replay keeps the restore-masked action (model.py:232)
                action, action_pol, _, _, hid = m.get_actions(self.obs, status="train", exploration=True,
                                                              actions_avail=self.avail, target=False, last_hid=self.hid)
                action_pol = action_pol.expand(N, m.n_, m.act_dim)
            env.step(m.env_action(action), fuse_obs=True, auto_reset=True)
            self._pack(action_pol.to(th.float32).contiguous(), hid.reshape(N, m.n_, -1).to(th.float32).contiguous())

    def step(self):
        """One vector step: a graph replay (or the eager body before capture) plus the host mirror of the ring cursor.
        Returns the physical slab the step completed."""
        if self.graph is not None:
            self.graph.replay()
        else:
            self.body()
        return self.buf.stepped()

    BURSTS = (16, 8, 4, 2)
    BURST_MAX = 96                                       # fused burst launch: steps per launch (an episode of the env)

    def run(self, m):
        """``m`` consecutive vector steps with nothing for the host to do in between: graphs of 16 / 8 / 4 / 2 bodies
        (captured on first use, sharing the one-step graph's memory pool) and single steps for the rest.  Every body
        finds its slabs through the device-side cursors, so k bodies in one graph are k replays of the one-step graph
        minus k - 1 graph launches (~8 us of idle GPU each at 4096 envs).  Returns the slabs completed, in order."""
        done = []
        for k in (self.burst_chunks(m) if self.graph is not None else [1] * m):
            if k == 1:
                done.append(self.step())
                continue
            g = self.bursts.get((k, self.fused_burst))      # (keyed by the body's form too: a flag flipped after capture()
            if g is None:                                    #  must not replay a graph recorded for the other body)
                g = self._capture_burst(k)
            g.replay()
            done.extend(self.buf.stepped() for _ in range(k))
        return done

    def burst_chunks(self, m):
        """The graph sizes run(m) replays, in order (1 = the one-step graph)."""
        out = []
        while m > 0:
            if self.fused_burst:
                k = max(1, min(m, self.BURST_MAX, self.buf.slabs - 1))     # (a burst stays below the ring's size)
            else:
                k = next((b for b in self.BURSTS if b <= m), 1)
            out.append(k)
            m -= k
        return out

    def _capture_burst(self, k, collect=True):
        """A graph of ``k`` bodies, sharing the one-step graph's memory pool.  Recording launches nothing: the environment,
        the ring and the cursors are untouched (the bodies were warmed up when the one-step graph was captured); only the
        env's host-side call counter would move."""
        calls = getattr(self.env, "calls", None)
        g = th.cuda.CUDAGraph()
        with graph_capture(g, collect=collect, pool=self.graph.pool()):
            if self.fused_burst:
                self.body(burst=k)
            else:
                for _ in range(k):
                    self.body()
        if calls is not None:
            self.env.calls = calls
        self.bursts[(k, self.fused_burst)] = g
        return g

    def capture(self, lengths=None):
        """Warm-up + capture.  The warm-up steps are real steps of the environment; the ring cursor and the statistics
        they moved are put back afterwards (the slabs they wrote are overwritten by the steps that follow).
        ``lengths``: the run lengths the caller will ask run() for (the training loop knows its schedule: the distances
        between update events and episode ends) — with the one-launch burst every length is a graph of its own, and they are
        all recorded NOW rather than inside somebody's timed region; other lengths are recorded at first use."""
        import os
        if not self.packable:
            raise RuntimeError("observation / hidden sizes outside flexnet_rollout_pack: the statistics would go through "
                               "ATen reductions, which must not be captured into a HIP graph (DESIGN.md §6)")
        cursor0 = self.buf.cursor.clone()
        if os.environ.get("FLEX_GRAPH_AUDIT") == "1":
            from .util import audit_graph_body
            self.audit = audit_graph_body(self.body)
        side = th.cuda.Stream()
        side.wait_stream(th.cuda.current_stream())
        with th.cuda.stream(side):
            for _ in range(3):
                self.body()                      # warm-up on a side stream, as graph capture requires
        th.cuda.current_stream().wait_stream(side)
        g = th.cuda.CUDAGraph()
        with graph_capture(g):
            self.body()
        self.graph = g
        self.bursts = {}
        # every burst size NOW, not at first use: a first use falls into somebody's timed region (a 16-body capture is
        # milliseconds of host work — BENCH_r02's SAFEMADDPG leg, timed from the second episode on, carried the 8 / 4 / 2
        # captures and read 0.366 ms per vector step on the driver's box against 0.25-0.32 on others)
        if self.fused_burst:
            sizes = sorted({k for m in (lengths or ()) for k in self.burst_chunks(int(m)) if k > 1})
        else:
            sizes = list(self.BURSTS)
        import gc
        gc.collect()
        for k in sizes:
            self._capture_burst(k, collect=False)
        self.buf.cursor.copy_(cursor0)

    def release(self):
        """Undo what construction set up on the env and the replay buffer (capture failed: the eager rollout takes over and
        writes the buffer field by field, so the env's device-side hooks must not point at a ring any more)."""
        env = self.env
        if self.cursor_stepped:
            env.set_step_counter(None)
            if hasattr(env, "set_obs_ring"):
                env.set_obs_ring(None, 0, 0)
            if hasattr(env, "set_replay_sink"):
                env.set_replay_sink(None, None, None, None, None)
        self.graph, self.bursts = None, {}
        self.buf.release_slabs()

    def start_episode(self, first_obs=None):
        """``first_obs`` given: a hard reset happened (env.reset()); the stream restarts in the ring and the hidden state
        is zero.  None: the environments continue where the last step left them (those that terminated restarted
        inside that launch, their hidden state was zeroed by the pack kernel)."""
        if first_obs is not None:
            if not self.ring_active and first_obs.data_ptr() != self._obs.data_ptr():
                self._obs.copy_(first_obs)
            self._hid.zero_()
            self.buf.begin_stream(first_obs)
        self._info_sum.zero_(); self._rew_sum.zero_(); self._fail_sum.zero_()
        if self.sink or self.summed_sink:
            self.acc.zero_()


class Model(nn.Module):
    """model.py:10-323, the parts the MADDPG path uses."""

    def __init__(self, args):
        super().__init__()
        self.args = args
        self.device = th.device("cuda" if th.cuda.is_available() and self.args.cuda else "cpu")
        self.n_ = self.args.agent_num
        self.hid_dim = self.args.hid_size
        self.obs_dim = self.args.obs_size
        self.act_dim = self.args.action_dim
        self.Transition = Transition
        self.batchnorm = nn.BatchNorm1d(self.n_)
        self.fused_inference = True      # no-grad policy passes on the GPU go through csrc/actor.hip

    # -- construction ------------------------------------------------------------------------------
    def construct_policy_net(self):
        """model.py:145-169"""
        input_shape = self.obs_dim + self.n_ if self.args.agent_id else self.obs_dim
        if self.args.gaussian_policy:
            raise NotImplementedError("gaussian policies are outside the MADDPG hot path")
        Agent = {"mlp": MLPAgent, "rnn": RNNAgent}[self.args.agent_type]
        count = 1 if self.args.shared_params else self.n_
        self.policy_dicts = nn.ModuleList([Agent(input_shape, self.args) for _ in range(count)])

    def init_weights(self, m):
        """model.py:174-182"""
        if type(m) == nn.Linear:
            if self.args.init_type == "normal":
                nn.init.normal_(m.weight, 0.0, self.args.init_std)
            elif self.args.init_type == "orthogonal":
                nn.init.orthogonal_(m.weight, gain=nn.init.calculate_gain(self.args.hid_activation))

    def reload_params_to_target(self):
        """model.py:22-26"""
        self.target_net.policy_dicts.load_state_dict(self.policy_dicts.state_dict())
        self.target_net.value_dicts.load_state_dict(self.value_dicts.state_dict())

    def update_target(self):
        """model.py:28-38: theta' <- (1 - target_lr) theta' + target_lr theta over every state_dict entry
        of the policy and value nets (floating entries; the nets hold no buffers)."""
        with th.no_grad():
            tgt = list(self.target_net.policy_dicts.state_dict().values()) + \
                list(self.target_net.value_dicts.state_dict().values())
            src = list(self.policy_dicts.state_dict().values()) + list(self.value_dicts.state_dict().values())
            th._foreach_mul_(tgt, 1 - self.args.target_lr)
            th._foreach_add_(tgt, src, alpha=self.args.target_lr)

    # -- policy ----------------------------------------------------------------------------------------
    def policy(self, obs, schedule=None, last_act=None, last_hid=None, info={}, stat={}):
        """model.py:102-140. obs [b, n, o] -> means [b, n, a], log_stds, hiddens [b, n, hid]."""
        b = obs.size(0)
        fused = None
        if self.args.shared_params and obs.is_cuda and not th.is_grad_enabled() and self.fused_inference:
            # rollout steps and bootstrap targets: one HIP launch instead of the module's ten kernels
            fused = fused_actor_forward(self.policy_dicts[0], obs, last_hid, self.n_, self.args.agent_id)
        if fused is not None:
            means = fused[0].view(b, self.n_, -1)
            hiddens = fused[1].view(b, self.n_, -1)
            return means, self._log_stds_like(means), hiddens
        if (self.args.shared_params and obs.is_cuda and th.is_grad_enabled() and self.fused_inference
                and b * self.n_ >= WGRAD_MIN_ROWS and isinstance(self.policy_dicts[0], RNNAgent)):
            # update batches: no id concat (the id block of fc1 is an addend per agent), fused LayerNorm/ReLU pass
            out = self.policy_dicts[0].forward_update(obs.reshape(b * self.n_, -1), last_hid, self.n_, self.args.agent_id)
            if out is not None:
                means, hiddens = out[0].view(b, self.n_, -1), out[2].view(b, self.n_, -1)
                return means, self._log_stds_like(means), hiddens
        if self.args.agent_id:
            ids = th.eye(self.n_, device=obs.device, dtype=obs.dtype).expand(b, -1, -1)
            obs = th.cat((obs, ids), dim=-1)
        if self.args.shared_params:
            means, _, hiddens = self.policy_dicts[0](obs.reshape(b * self.n_, -1), last_hid)
            means = means.view(b, self.n_, -1)
            hiddens = hiddens.view(b, self.n_, -1)
        else:
            outs = [pol(obs[:, i, :], last_hid[:, i, :]) for i, pol in enumerate(self.policy_dicts)]
            means = th.stack([o[0] for o in outs], dim=1)
            hiddens = th.stack([o[2] for o in outs], dim=1)
        # fixed_policy_std (model.py:121-123): log(1.0) = 0 at the default config
        return means, self._log_stds_like(means), hiddens

    def _log_stds_like(self, means):
        """fixed_policy_std (model.py:121-123) as a broadcast view of ONE cached element per device — a fill kernel per
        policy call otherwise; nothing downstream writes into it."""
        cache = self.__dict__.setdefault("_log_std_cache", {})
        if means.device not in cache:
            log_std = float(np.log(self.args.fixed_policy_std))
            # the entropy of N(mean, std) does not depend on the mean: with the fixed std it is ONE number
            cache[means.device] = (th.full((1,), log_std, dtype=means.dtype, device=means.device),
                                   th.full((), 0.5 + 0.5 * float(np.log(2.0 * np.pi)) + log_std, dtype=means.dtype,
                                           device=means.device))
        out = cache[means.device][0].expand_as(means)
        out._flex_entropy = cache[means.device][1]        # mean entropy of Normal(means, exp(out)) (util.py:35-36)
        return out

    # -- update cadence (model.py:40-71) -----------------------------------------------------------
    def transition_update(self, trainer, trans, stat):
        if self.args.replay:
            if trans is not None:
                trainer.replay_buffer.add_experience(trans)
            replay_cond = trainer.steps > self.args.replay_warmup \
                and len(trainer.replay_buffer.buffer) >= trainer.effective_batch_size() \
                and trainer.steps % self.args.behaviour_update_freq == 0
            if replay_cond:
                if hasattr(trainer, "replay_event"):      # same sub-updates in the same order, software-pipelined
                    trainer.replay_event(stat, self.args.value_update_epochs, self.args.policy_update_epochs)
                else:
                    for _ in range(self.args.value_update_epochs):
                        trainer.value_replay_process(stat)
                    for _ in range(self.args.policy_update_epochs):
                        trainer.policy_replay_process(stat)
        else:
            raise NotImplementedError("the MADDPG path always replays (default.yaml:22)")
        if self.args.target and trainer.steps % self.args.target_update_freq == 0:
            self.update_target()

    # -- batches -----------------------------------------------------------------------------------------
    def unpack_data(self, batch, normalise_reward=True):
        """model.py:308-323.  ``batch`` is a Transition of device tensors (replay_buffer.window) or, for
        callers written against the reference, a Transition of per-sample tuples (trainer.py:68).
        ``normalise_reward=False`` hands the raw reward on: the caller applies the BatchNorm itself (the fused value
        loss does, together with the running-statistics update — exactly once per get_loss call either way)."""
        if not isinstance(batch.reward, th.Tensor):
            batch = Transition(*[th.stack([th.as_tensor(np.asarray(x), dtype=th.float32) for x in f]).to(self.device)
                                 for f in batch])
            # model.py:312-320 concatenates the [1, n, x] per-step arrays along axis 0
            batch = Transition(*[f.squeeze(1) if f.dim() == 4 else f for f in batch])
        reward = batch.reward.float()
        if self.args.reward_normalisation and normalise_reward:
            # train-mode batch statistics (running stats still update).  The affine pair is in no optimiser
            # (trainer.py:34-35), so its gradient is never used: taking it out of the graph removes a
            # batch-norm backward over [batch, n] that cost 31 % of a value sub-update.
            # The reward columns of a packed replay row are a strided slice: the statistics kernel reads them 3x
            # slower than a contiguous copy (41 vs 14 us for [32768, 5]), so copy first.
            with th.no_grad():
                reward = sync_batchnorm(self.batchnorm, reward.contiguous())     # (= self.batchnorm(reward) on one rank)
        done = batch.done.float().view(-1, 1)
        last_step = batch.last_step.float().view(-1, 1)
        # model.py:313 fills log_prob_a from batch.action; nothing downstream reads it
        return (batch.state, batch.action, batch.action, batch.value, batch.next_value, reward, batch.next_state,
                done, last_step, batch.action_avail, batch.last_hid, batch.hid)

    # -- rollouts ------------------------------------------------------------------------------------------
    def train_process(self, stat, trainer):
        if hasattr(trainer.env, "handle"):          # VecFlexProvisionEnv: batched, sync-free rollout
            return self._train_process_vec(stat, trainer)
        return self._train_process_single(stat, trainer)

    def _train_process_single(self, stat, trainer):
        """model.py:198-267 on a reference-style env (lists of numpy observations, one env)."""
        env = trainer.env
        stat_train = {"mean_train_reward": 0}
        state, _ = env.reset()
        last_hid = self.policy_dicts[0].init_hidden()
        avail = th.tensor(env.get_avail_actions())
        t = 0
        for t in range(self.args.max_steps):
            state_ = prep_obs(state).to(self.device).contiguous().view(1, self.n_, self.obs_dim)
            action, action_pol, log_prob_a, _, hid = self.get_actions(state_, status="train", exploration=True,
                                                                      actions_avail=avail, target=False, last_hid=last_hid)
            value = self.value(state_, action_pol)
            _, actual = translate_action(self.args, action, env)
            reward, done, info = env.step(actual)
            next_state = env.get_obs()
            next_state_ = prep_obs(next_state).to(self.device).contiguous().view(1, self.n_, self.obs_dim)
            _, next_action_pol, _, _, _ = self.get_actions(next_state_, status="train", exploration=True,
                                                           actions_avail=avail, target=False, last_hid=hid)
            next_value = self.value(next_state_, next_action_pol)
            done_ = done or t == self.args.max_steps - 1
            trans = Transition(state, action_pol.detach().cpu().numpy(), log_prob_a.detach().cpu().numpy(),
                               value.detach().cpu().numpy(), next_value.detach().cpu().numpy(),
                               np.array([reward] * env.get_num_of_agents()), next_state, done, done_,
                               env.get_avail_actions(), last_hid.detach().cpu().numpy(), hid.detach().cpu().numpy())
            self.transition_update(trainer, trans, stat)
            for k, v in info.items():
                stat_train["mean_train_" + k] = stat_train.get("mean_train_" + k, 0) + v
            stat_train["mean_train_reward"] += reward
            trainer.steps += 1
            if done_:
                break
            state, last_hid = next_state, hid
        trainer.episodes += 1
        for k, v in stat_train.items():
            if k.split("_")[0] == "mean":
                stat_train[k] = v / float(t + 1)
        stat.update(stat_train)

    def _train_process_vec(self, stat, trainer):
        """The same rollout over N environments at once.  One "episode" is ``episode_limit - 1`` vector
        steps (what one reference episode executes); environments that terminate early
        (solver failure) are auto-reset by mask.  Nothing in the loop reads a device value on the host;
        statistics are accumulated on the device and fetched once at the end."""
        env = trainer.env
        args = self.args
        N = env.n_envs
        horizon = min(args.max_steps, env.episode_limit - 1)
        if self._use_rollout_graph(trainer):
            return self._train_process_graph(stat, trainer, horizon)
        obs = env.reset().clone()
        last_hid = th.zeros(N, self.n_, self.hid_dim, device=self.device)
        info_sum = th.zeros(env.info.shape[1], dtype=th.float64, device=self.device)
        rew_sum = th.zeros((), dtype=th.float64, device=self.device)
        fail_sum = th.zeros((), dtype=th.float64, device=self.device)
        buf = trainer.replay_buffer
        buf.const_shapes = {"log_prob_a": (self.n_, self.act_dim), "value": (self.n_, 1), "next_value": (self.n_, 1),
                            "action_avail": (self.n_, self.act_dim)}
        std = float(self.args.fixed_policy_std)
        for t in range(horizon):
            with th.no_grad():
                # exploration of select_action (util.py:57-64): tanh(N(mean, std)); every action is available
                # (env:721-730), so the restore mask is the identity and log_prob — unused by the DDPG losses — is
                # not formed.  SAFEMADDPG routes through get_actions for its safety layer.
                if type(self).get_actions is MADDPG.get_actions and self.args.action_enforcebound:
                    means, _, hid = self.policy(obs, last_hid=last_hid)
                    action = action_pol = th.tanh(means + std * th.randn_like(means))
                else:
                    avail = th.ones(N, self.n_, self.act_dim, device=self.device)
                    action, action_pol, _, _, hid = self.get_actions(obs, status="train", exploration=True,
                                                                     actions_avail=avail, target=False, last_hid=last_hid)
                actual = self.env_action(action)
            # one launch: step + get_obs, and envs that terminate restart in place (their row of env.obs is then
            # the first observation of the new episode; the terminal transition is masked by `done` in the loss)
            reward, done, info = env.step(actual, fuse_obs=True, auto_reset=True)
            next_obs = env.obs
            donef = done.float()
            last_step = donef if t < horizon - 1 else th.ones_like(donef)       # model.py:229
            buf.add_batch(state=obs, action=action_pol, log_prob_a=0.0, value=0.0, next_value=0.0,
                          reward=reward.float().unsqueeze(1).expand(N, self.n_), next_state=next_obs, done=donef,
                          last_step=last_step, action_avail=1.0, last_hid=last_hid, hid=hid)
            self.transition_update(trainer, None, stat)
            info_sum += info.sum(0)
            rew_sum += reward.sum()
            fail_sum += env.failed.sum()
            trainer.steps += 1
            # next iteration: terminated envs have restarted; they get a zero hidden state
            obs = next_obs.clone()
            last_hid = hid * (1.0 - donef).view(N, 1, 1)
        trainer.episodes += 1
        denom = float(N * horizon)
        vals = th.cat([info_sum, rew_sum.view(1), fail_sum.view(1)]).cpu().numpy() / denom
        from ._lib import INFO_KEYS
        for i, k in enumerate(INFO_KEYS):
            stat["mean_train_" + k] = float(vals[i])
        stat["mean_train_reward"] = float(vals[-2])
        stat["mean_train_solver_failed"] = float(vals[-1])

    def _use_rollout_graph(self, trainer):
        """HIP-graph rollout: CUDA device, not switched off by the trainer.  MADDPG and SAFEMADDPG take the fused two-kernel
        body, MATD3 and IDDPG the general body around their own get_actions; if capture fails the eager loop runs."""
        if not getattr(trainer, "graph_rollout", True) or self.device.type != "cuda":
            return False
        if type(self).__name__ == "SAFEMADDPG":          # the graph body knows the reference's safety-layer order only
            return type(self).get_actions is SAFEMADDPG.get_actions and bool(self.args.action_enforcebound)
        return True                                      # MADDPG: fused path; MATD3 / IDDPG / others: their get_actions

    def _run_lengths(self, steps, horizon):
        """Every run length the loop of _train_process_graph will ask the rollout for, from step counter ``steps`` on: the
        schedule repeats once the counter's phase against the update frequencies repeats at an episode start."""
        freqs = [int(self.args.behaviour_update_freq)] + ([int(self.args.target_update_freq)] if self.args.target else [])
        freqs = [f for f in freqs if f > 0]
        period = int(np.lcm.reduce(freqs)) if freqs else 1
        seen, lengths, s = set(), set(), int(steps)
        while s % period not in seen and len(seen) < 4096:
            seen.add(s % period)
            t = 0
            while t < horizon:
                m = min([horizon - t] + [(-s) % f + 1 for f in freqs])
                lengths.add(m)
                t += m
                s += m
        return sorted(lengths)

    def _train_process_graph(self, stat, trainer, horizon):
        env, buf = trainer.env, trainer.replay_buffer
        N = env.n_envs
        rg = getattr(self, "_rollout_graph", None)
        if rg is None or rg.env is not env or rg.buf is not buf:
            rg = None
            try:
                rg = RolloutGraph(self, env, buf)
                rg.start_episode(env.reset())             # the warm-up steps of capture() need a live episode
                rg.capture(lengths=self._run_lengths(trainer.steps, horizon))
            except Exception as exc:                      # capture unsupported here: fall back to the eager loop
                import warnings
                warnings.warn(f"rollout graph capture failed ({exc}); using the eager rollout")
                trainer.graph_rollout = False
                if rg is not None:
                    rg.release()                          # env hooks off, slab ring dropped: add_batch works again
                elif getattr(buf, "slab_mode", False) and buf.length == 0:
                    buf.release_slabs()
                return self._train_process_vec(stat, trainer)
            object.__setattr__(self, "_rollout_graph", rg)
        # model.py:208 resets the environment at the start of every episode.  Here an episode of `horizon` vector steps
        # ends with every environment terminated and restarted IN the last launch (fresh draws from the same reset
        # stream), so the next episode continues from that state: same distribution, and the replay ring stays one
        # unbroken stream (next_state of slab k is slab k + 1).  A hard reset happens when that is not the case: the
        # first episode, environments out of phase (a solver failure restarted some mid-episode), or anybody else having
        # stepped / reset the env since (evaluation).
        if rg.env_calls == getattr(env, "calls", None) and getattr(rg, "in_phase", False):
            rg.start_episode(None)
        else:
            rg.start_episode(env.reset())
        small = buf.small_ring
        last_col = self.n_ * self.act_dim + self.n_ + 1
        # model.py:215-262 steps, then asks transition_update whether an update is due — which it is only when the step
        # counter reaches a multiple of behaviour_update_freq / target_update_freq (model.py:43-50).  The steps up to the
        # next such multiple (or the end of the episode) need nothing from the host and go out as bursts.
        freqs = [int(self.args.behaviour_update_freq)] + ([int(self.args.target_update_freq)] if self.args.target else [])
        t = 0
        while t < horizon:
            s0 = trainer.steps
            m = min([horizon - t] + [(-s0) % f + 1 for f in freqs if f > 0])
            slabs = rg.run(m)
            t += m
            if t == horizon:                                                  # model.py:229: last_step on the final step
                small[slabs[-1], :, last_col] = 1.0
            trainer.steps = s0 + m - 1
            self.transition_update(trainer, None, stat)
            trainer.steps += 1
        trainer.episodes += 1
        vals = th.cat([rg.info_sum, rg.rew_sum.view(1), rg.fail_sum.view(1), env.done.double().sum().view(1)]).cpu().numpy()
        rg.in_phase = bool(vals[-1] == N)                 # everybody terminated (and restarted) in the last launch
        rg.env_calls = getattr(env, "calls", None)
        vals = vals[:-1] / float(N * horizon)
        from ._lib import INFO_KEYS
        for i, k in enumerate(INFO_KEYS):
            stat["mean_train_" + k] = float(vals[i])
        stat["mean_train_reward"] = float(vals[-2])
        stat["mean_train_solver_failed"] = float(vals[-1])

    def env_action(self, action):
        """translate_action (util.py:121-130) kept on the device: [N, n, a] in the env's range.  MATD3 samples ONE
        action from the agent-summed mean (matd3.py:92-97) — in the reference that [1,1,a] tensor cannot even be
        reshaped by env:260; here every agent receives it, which is what its restore_actions broadcast implies."""
        action = action.detach()
        if action.dim() == 3 and action.size(1) == 1 and self.n_ > 1:
            action = action.expand(-1, self.n_, -1)
        return scale_action(self.args, action).reshape(-1, self.n_, self.act_dim)

    def evaluation(self, stat, trainer):
        """model.py:269-306 (test-mode rollouts; next-row f1 of SURVEY §8f)."""
        env = trainer.env
        if hasattr(env, "handle"):
            return self._evaluation_vec(stat, trainer)
        stat_test = {}
        for _ in range(self.args.num_eval_episodes):
            epi = {"mean_test_reward": 0}
            state, _ = env.reset()
            last_hid = self.policy_dicts[0].init_hidden()
            avail = th.tensor(env.get_avail_actions())
            t = 0
            for t in range(self.args.max_steps):
                state_ = prep_obs(state).to(self.device).contiguous().view(1, self.n_, self.obs_dim)
                with th.no_grad():
                    action, _, _, _, hid = self.get_actions(state_, status="test", exploration=False,
                                                            actions_avail=avail, target=False, last_hid=last_hid)
                _, actual = translate_action(self.args, action, env)
                reward, done, info = env.step(actual)
                done_ = done or t == self.args.max_steps - 1
                next_state = env.get_obs()
                for k, v in info.items():
                    epi["mean_test_" + k] = epi.get("mean_test_" + k, 0) + v
                epi["mean_test_reward"] += reward
                if done_:
                    break
                state, last_hid = next_state, hid
            for k, v in epi.items():
                stat_test[k] = stat_test.get(k, 0) + v / float(t + 1)
        for k, v in stat_test.items():
            stat[k] = v / float(self.args.num_eval_episodes)

    def _evaluation_vec(self, stat, trainer):
        stat.update(self.evaluate_on(trainer.env))

    def evaluate_on(self, env, spec=None):
        """model.py:269-306 on every environment of ``env`` (a VecFlexProvisionEnv, not necessarily the trainer's): one
        test-mode episode each — tanh(mean), no exploration (util.py:79-82) — and the reference's ``mean_test_*`` keys,
        averaged over environments and steps.  ``spec``: injected episode draws (VecFlexProvisionEnv.reset), so that
        different policies are compared on the SAME episodes.  Besides the reference's keys: ``mean_test_violation_rate``
        = share of env-steps with any bus voltage outside [v_min, v_max] (voltage_penalty > 0, env:685) and
        ``mean_test_solver_failed``."""
        bound = self.__dict__.get("env", None)
        rebind = bound is not None and (bound.vec if hasattr(bound, "vec") else bound) is not env
        if rebind:                    # SAFEMADDPG projects against the env it is bound to (safemaddpg.py:143-172)
            self.__dict__["env"] = env
        try:
            return self._evaluate_on(env, spec)
        finally:
            if rebind:
                self.__dict__["env"] = bound

    def _evaluate_on(self, env, spec):
        N = env.n_envs
        horizon = min(self.args.max_steps, env.episode_limit - 1)
        obs = env.reset(spec=spec).clone()
        viol_sum = th.zeros((), dtype=th.float64, device=self.device)
        fail_sum = th.zeros((), dtype=th.float64, device=self.device)
        last_hid = th.zeros(N, self.n_, self.hid_dim, device=self.device)
        avail = th.ones(N, self.n_, self.act_dim, device=self.device)
        info_sum = th.zeros(env.info.shape[1], dtype=th.float64, device=self.device)
        rew_sum = th.zeros((), dtype=th.float64, device=self.device)
        # plain MADDPG in test mode is tanh(mean) (util.py:79-82) -> translate_action: the actor kernel's epilogue with a zero
        # standard deviation, one launch instead of the policy plus a dozen pointwise ones
        fused = (type(self).get_actions is MADDPG.get_actions and self.fused_eval and self.fused_inference
                 and self.args.shared_params and env.obs.is_cuda and self.args.agent_type == "rnn" and self.hid_dim == 64
                 and self.obs_dim <= 144 and bool(self.args.action_enforcebound))
        zero_noise = th.zeros(N, self.n_, self.act_dim, device=self.device) if fused else None
        for t in range(horizon):
            with th.no_grad():
                out = fused_actor_forward(self.policy_dicts[0], obs, last_hid, self.n_, self.args.agent_id, noise=zero_noise,
                                          std=0.0, low=self.args.action_low, high=self.args.action_high) if fused else None
                if out is not None:
                    hid, actual = out[1].view(N, self.n_, -1), out[3].view(N, self.n_, self.act_dim)
                else:
                    action, _, _, _, hid = self.get_actions(obs, status="test", exploration=False, actions_avail=avail,
                                                            target=False, last_hid=last_hid)
                    actual = self.env_action(action)
            reward, done, info = env.step(actual, fuse_obs=True, auto_reset=True)
            info_sum += info.sum(0)
            rew_sum += reward.sum()
            viol_sum += (info[:, 5] > 0).sum()
            fail_sum += env.failed.sum()
            donef = done.float()
            obs = env.obs.clone()
            last_hid = hid * (1.0 - donef).view(N, 1, 1)
        vals = th.cat([info_sum, rew_sum.view(1), viol_sum.view(1), fail_sum.view(1)]).cpu().numpy() / float(N * horizon)
        from ._lib import INFO_KEYS
        stat = {}
        for i, k in enumerate(INFO_KEYS):
            stat["mean_test_" + k] = float(vals[i])
        stat["mean_test_reward"] = float(vals[-3])
        stat["mean_test_violation_rate"] = float(vals[-2])
        stat["mean_test_solver_failed"] = float(vals[-1])
        return stat


class MADDPG(Model):
    """maddpg.py:7-123."""

    def __init__(self, args, target_net=None):
        super().__init__(args)
        self.construct_model()
        self.apply(self.init_weights)
        if target_net is not None:
            self.target_net = target_net
            self.reload_params_to_target()
        self.batchnorm = nn.BatchNorm1d(self.args.agent_num).to(self.device)      # maddpg.py:16

    def construct_value_net(self):
        """maddpg.py:18-27: input (obs+act)*n + n."""
        input_shape = (self.obs_dim + self.act_dim) * self.n_ + (self.n_ if self.args.agent_id else 0)
        count = 1 if self.args.shared_params else self.n_
        self.value_dicts = nn.ModuleList([MLPCritic(input_shape, 1, self.args) for _ in range(count)])

    # replay fields each loss reads (get_loss below; MATD3 and SAFEMADDPG read the same ones): a graphed sub-update
    # refreshes only these columns of its static batch
    # (reward in both: unpack_data's batch-norm running statistics move on every get_loss call, model.py:308-323)
    graph_safe_updates = True       # trainer._graphed_sub_update: the gradient path uses no multi-block PyTorch reduction
    update_fields = {"policy": ("state", "reward", "last_hid"),
                     "value": ("state", "action", "reward", "next_state", "done", "hid"),
                     # trainer.replay_event with the bootstrap values filed per transition first (round 3)
                     "value_cached": ("state", "action", "reward", "done", "next_value"),
                     "bootstrap": ("next_state", "hid")}

    def construct_model(self):
        self.construct_value_net()
        self.construct_policy_net()

    def value(self, obs, act, critic_frozen=False):
        """maddpg.py:33-76.  Row i of the reference's critic input is
        [obs_0..obs_{n-1} | onehot(i) | act_0..act_{n-1}] with act_j detached for j != i.  fc1 of that row is
        W_obs @ obs_all + W_id[:, i] + W_act @ act_all(detached) + W_act[:, block i] @ (act_i - act_i.detach()) + b:
        the first and third terms are shared by the n rows of a sample, the last one is zero-valued and only
        carries agent i's own-action gradient.  Returns [b, n, 1].  ``critic_frozen``: the caller differentiates w.r.t. the
        actions only (the policy loss of a policy sub-update) — one autograd node, no critic-parameter gradients."""
        b, n, o, a = obs.size(0), self.n_, self.obs_dim, self.act_dim
        if (critic_frozen and self.args.shared_params and self.args.agent_id and act.requires_grad and th.is_grad_enabled()
                and critic_policy_supported(self.value_dicts[0], obs.reshape(b, n * o), act, n)):
            return CriticTail.apply_policy(obs.reshape(b, n * o), act, self.value_dicts[0]).view(b, n, 1)
        act_det = act.detach()
        own = act - act_det if act.requires_grad else None                # zeros that carry d/d act_i
        values = []
        nets = self.value_dicts if not self.args.shared_params else [self.value_dicts[0]] * 1
        if self.args.shared_params:
            net = self.value_dicts[0]
            W, bias = net.fc1.weight, net.fc1.bias
            W_obs = W[:, :n * o]
            off = n * o
            if self.args.agent_id:
                W_id = W[:, off:off + n]                                  # [hid, n]
                off += n
            W_act = W[:, off:off + n * a]
            act_cols = act_det.reshape(b, n * a)
            obs_cols = obs.reshape(b, n * o)
            if (not act.requires_grad and self.args.agent_id and th.is_grad_enabled() and W.requires_grad
                    and critic_replayed_supported(net, obs_cols, act_cols, n)):
                # value loss on replayed actions: the whole critic as one autograd node (nets._CriticReplayedFn)
                return CriticTail.apply_replayed(obs_cols, act_cols, n, net).view(b, n, 1)
            if not th.is_grad_enabled() or not (W.requires_grad or act.requires_grad or obs.requires_grad):
                # bootstrap targets: no graph — one launch of csrc/linear.hip at update sizes (nets.critic_first_layer), else
                # two library GEMMs (the bias rides the first, the second accumulates)
                with th.no_grad():
                    shared = critic_first_layer(bias, obs_cols, act_cols, W, off)
            else:
                shared = wide_batch_linear(obs_cols, W_obs) + wide_batch_linear(act_cols, W_act) + bias      # [b, hid]
            if not act.requires_grad and self.args.agent_id and critic_tail_supported(net, shared):
                # replayed actions (value loss, bootstrap target): every row is shared[b] + the agent's id column; the
                # fused tail composes it on the fly instead of reading a materialised [b * n, hid] tensor
                return CriticTail.apply_composed(shared, W_id.t(), net).view(b, n, 1)
            h = shared.unsqueeze(1).expand(b, n, -1)
            if self.args.agent_id:
                h = h + W_id.t().unsqueeze(0)                             # [1, n, hid]
            if act.requires_grad:      # zero-valued term that only carries d/d act_i: nothing to add for replayed actions
                h = h + th.einsum("bia,hia->bih", own, W_act.view(-1, n, a))
            v, _ = net.forward_from_hidden(h.reshape(b * n, -1), need_hidden=False)
            return v.view(b, n, 1)
        for i, net in enumerate(nets):                                    # non-shared critics: plain input rows
            acts_i = act_det.clone()
            acts_i[:, i] = act[:, i]
            ids = th.eye(n, device=obs.device, dtype=obs.dtype)[i].expand(b, -1)
            parts = [obs.reshape(b, n * o)] + ([ids] if self.args.agent_id else []) + [acts_i.reshape(b, n * a)]
            v, _ = net(th.cat(parts, dim=-1), None)
            values.append(v)
        return th.stack(values, dim=1)

    def get_actions(self, state, status, exploration, actions_avail, target=False, last_hid=None):
        """maddpg.py:78-98 (continuous branch)."""
        pol = self.target_net.policy if (target and self.args.target) else self.policy
        means, log_stds, hiddens = pol(state, last_hid=last_hid)
        actions, log_prob_a = select_action(self.args, means, status=status, exploration=exploration,
                                            info={"log_std": log_stds})
        if getattr(actions_avail, "_flex_const", None) == 1.0:      # every action available (env:721-730): the mask is 1
            restore_actions = actions
        else:
            restore_mask = 1.0 - (actions_avail.to(means.device) == 0).float()
            restore_actions = restore_mask * actions
        return actions, restore_actions, log_prob_a, (means, log_stds), hiddens

    bootstrap_from_batch = False     # set by trainer.replay_event while it captures / replays value sub-updates on filed values
    bootstrap_cacheable = True       # get_loss below honours bootstrap_from_batch (a subclass with its own get_loss says False)

    def bootstrap_values(self, next_state, actions_avail, hids):
        """Q'(s', pi(s')) [b, n] of maddpg.py:108-111: the next action from the behaviour policy (double_q) or the target
        policy, valued by the target critic; no gradient (maddpg.py:110,115: .detach())."""
        with th.no_grad():
            _, next_actions, _, _, _ = self.get_actions(next_state, status="train", exploration=False,
                                                        actions_avail=actions_avail, target=not self.args.double_q,
                                                        last_hid=hids)
            return self.target_net.value(next_state, next_actions).view(-1, self.n_)

    def get_loss(self, batch, need="both"):
        """maddpg.py:100-123: policy_loss = -Q(s, pi(s)).mean(); value_loss = (r + gamma (1-done) Q'(s', pi(s')) - Q(s,a))^2.mean().

        The reference evaluates both losses in every sub-update and uses one (trainer.py:84,101).  ``need`` =
        "value" / "policy" evaluates only the graph that loss needs — same loss value, same gradients, about half
        the forward work and no backward through the unused half; "both" is the reference's call."""
        # value loss alone on the GPU: reward normalisation, TD target, loss and dLoss/dQ are csrc/tdloss.hip
        on_gpu = (isinstance(batch.reward, th.Tensor) and batch.reward.is_cuda and self.fused_inference
                  and th.is_grad_enabled())
        fused_td = need == "value" and on_gpu
        # the policy loss never reads the reward: all the normalisation owes is the module's running statistics
        stats_only = (need == "policy" and on_gpu and self.args.reward_normalisation
                      and batchnorm_stats_supported(self.batchnorm, batch.reward))
        state, actions, _, _, _, rewards, next_state, done, _, actions_avail, last_hids, hids = \
            self.unpack_data(batch, normalise_reward=not (fused_td or stats_only))
        if stats_only:
            batchnorm_update_running_stats(self.batchnorm, rewards)
        policy_loss = value_loss = None
        action_out = None
        if need in ("both", "policy"):
            _, actions_pol, _, action_out, _ = self.get_actions(state, status="train", exploration=False,
                                                                actions_avail=actions_avail, target=False,
                                                                last_hid=last_hids)
            # need == "policy": a policy sub-update — only the policy optimiser's parameters take this loss's gradient
            # (utils/trainer.py:99-108), so the critic is differentiated w.r.t. the actions alone
            policy_loss = self._critic_policy_loss(state, actions_pol) if need == "policy" else None
            if policy_loss is None:
                advantages = self.value(state, actions_pol, critic_frozen=(need == "policy")).view(-1, self.n_)
                if self.args.normalize_advantages:
                    advantages = self.batchnorm(advantages)
                policy_loss = mean_all(advantages, sign=-1.0)
        if need in ("both", "value"):
            if self.bootstrap_from_batch and need == "value":
                # trainer.replay_event filed Q'(s', pi(s')) for this window already (bootstrap_values below, on the same
                # networks: they do not change between the value sub-updates of one update event)
                next_values = batch.next_value.view(-1, self.n_)
            else:
                next_values = self.bootstrap_values(next_state, actions_avail, hids)
            bn = self.batchnorm if self.args.reward_normalisation else None
            if fused_td:
                # update batches: Q(s, a), the TD error and the critic's whole backward in one pass (nets._CriticTdLossFn)
                value_loss = self._critic_td_loss(state, actions, next_values, rewards, done, bn)
                if value_loss is not None:
                    return policy_loss, value_loss, action_out
            values = self.value(state, actions).view(-1, self.n_)
            if fused_td and td_loss_supported(values, next_values, rewards, done, bn):
                value_loss = td_loss(values, next_values, rewards, done, self.args.gamma, bn)
            else:
                if fused_td and bn is not None:                 # the raw reward was handed on: normalise it here
                    with th.no_grad():
                        rewards = sync_batchnorm(bn, rewards.contiguous())
                returns = rewards + self.args.gamma * (1 - done) * next_values
                assert returns.size() == values.size()
                value_loss = mean_all((returns - values).pow(2))
        return policy_loss, value_loss, action_out


def _maddpg_critic_td_loss(self, state, actions, next_values, rewards, done, bn):
    """MADDPG.value + the value loss + the critic's backward as one node, or None where that node does not apply."""
    if not (self.args.shared_params and self.args.agent_id and not actions.requires_grad and self.fused_td_backward
            and type(self).value is MADDPG.value):          # (IDDPG / MATD3 bring their own critic input layout)
        return None
    b, n = state.size(0), self.n_
    net = self.value_dicts[0]
    obs_cols, act_cols = state.reshape(b, n * self.obs_dim), actions.reshape(b, n * self.act_dim)
    if not critic_td_loss_supported(net, obs_cols, act_cols, n, next_values, rewards, done, bn):
        return None
    return critic_td_loss(obs_cols, act_cols, n, net, next_values, rewards, done, self.args.gamma, bn)


def _maddpg_critic_policy_loss(self, state, actions_pol):
    """-mean Q(s, pi(s)) with the critic frozen (a policy sub-update) as one node, or None where that node does not apply."""
    if not (self.args.shared_params and self.args.agent_id and self.fused_td_backward and not self.args.normalize_advantages
            and type(self).value is MADDPG.value and isinstance(actions_pol, th.Tensor) and actions_pol.is_cuda
            and self.fused_inference):
        return None
    b, n = state.size(0), self.n_
    net = self.value_dicts[0]
    obs_cols = state.reshape(b, n * self.obs_dim)
    if not critic_policy_loss_supported(net, obs_cols, actions_pol, n):
        return None
    return critic_policy_loss(obs_cols, actions_pol, net, sign=-1.0)


def _maddpg_reads_state_in_place(self, bs):
    """True iff a value sub-update on filed bootstrap values ("value_cached") touches ``batch.state`` ONLY through
    nets._CriticTdLossFn — whose first layer (flexnet_linear2) and weight gradient (flexnet_wgrad) can read the window in place
    from the replay's stacked-observation ring (nets.RING_VIEWS; trainer._static_batch then hands a NaN placeholder instead of
    a gathered copy).  The conditions are those of _maddpg_critic_td_loss / nets.critic_td_loss_supported that do not depend on
    the batch's values; a subclass with its own get_loss / value / critic input layout (MATD3, IDDPG) answers False."""
    from .nets import CRITIC_TD_MIN_ROWS, CRITIC_VARIANT
    import safe_marl_amd.nets as _nets
    n = self.n_
    net = self.value_dicts[0]
    fc1 = getattr(net, "fc1", None)
    return bool(type(self).get_loss is MADDPG.get_loss and type(self)._critic_td_loss is _maddpg_critic_td_loss
                and type(self).value is MADDPG.value and self.args.shared_params and self.args.agent_id
                and self.fused_td_backward and self.fused_inference and _nets.CRITIC_FC1_FUSED
                and bs * n >= CRITIC_TD_MIN_ROWS and CRITIC_VARIANT == 0 and _nets.CRITIC_PGRAD32 == 0
                and (n * self.obs_dim) % 8 == 0 and (n * self.act_dim) % 4 == 0 and n * (self.obs_dim + self.act_dim) <= 768
                and fc1 is not None and fc1.weight.shape == (64, n * (self.obs_dim + self.act_dim) + n))


def _maddpg_reads_next_state_in_place(self, bs):
    """The same question for the graph that files the bootstrap values (MADDPG.bootstrap_values on "next_state", "hid"): the
    policy's fused inference pass (nets.fused_actor_forward) and the target critic's first layer without a graph
    (nets.critic_first_layer) are the only readers of ``next_state``."""
    import safe_marl_amd.nets as _nets
    n = self.n_
    tgt = getattr(self, "target_net", None)
    if tgt is None or type(self).bootstrap_values is not MADDPG.bootstrap_values or type(self).policy is not Model.policy:
        return False
    agent, net = self.policy_dicts[0], tgt.value_dicts[0]
    fc1 = getattr(net, "fc1", None)
    a = agent.args
    return bool(type(self).get_actions is MADDPG.get_actions and type(tgt).value is MADDPG.value and self.args.shared_params
                and self.args.agent_id and self.fused_inference and tgt.fused_inference and _nets.CRITIC_FC1_FUSED
                and isinstance(agent, _nets.RNNAgent) and a.hid_size == 64 and a.hid_activation == "relu"
                and self.obs_dim <= 144 and n <= 8 and a.action_dim <= 8
                and (n * self.obs_dim) % 8 == 0 and (n * self.act_dim) % 4 == 0 and n * (self.obs_dim + self.act_dim) <= 768
                and fc1 is not None and fc1.weight.shape == (64, n * (self.obs_dim + self.act_dim) + n)
                and bs * n >= 65536)


MADDPG.reads_state_in_place = _maddpg_reads_state_in_place
MADDPG.reads_next_state_in_place = _maddpg_reads_next_state_in_place
MADDPG._critic_td_loss = _maddpg_critic_td_loss
MADDPG._critic_policy_loss = _maddpg_critic_policy_loss
MADDPG.fused_eval = True                  # (tests switch it off to compare the evaluation with the tensor composition)
MADDPG.fused_td_backward = True          # (tests switch it off to compare with the forward / td_loss / backward sequence)


def summed_exploration(model, means, env_action=None, action_out=None):
    """tanh(sum over agents of the means + exp(sum of log-stds) * eps), the ONE action of matd3.py:92-97 / iddpg.py:66-71
    under util.py:57-64, handed to every agent: [b, n, a] (and, with ``env_action``, translate_action of it) from one launch
    of flexnet_agent_sum_explore.  eps is Normal.rsample's own draw; every fp32 rounding sits where the tensor ops have it."""
    import ctypes as C
    import torch.distributions.normal as tdn      # (looked up at call time: the very function Normal.rsample calls)
    from . import _lib
    b, n, a = means.shape
    cache = model.__dict__.setdefault("_std_sum_cache", {})
    if means.device not in cache:
        with th.no_grad():
            ls = model._log_stds_like(th.zeros(1, n, a, device=means.device))
            cache[means.device] = _sum_agents(ls.expand(1, n, a)).exp().reshape(a).to(th.float32).contiguous()
    means = means.contiguous()
    eps = tdn._standard_normal((b, 1, a), means.dtype, means.device)
    out = action_out if action_out is not None else th.empty(b, n, a, dtype=th.float32, device=means.device)
    k = _lib.FlexAgentSumArgs()
    k.n_envs, k.n_agents, k.act_dim = b, n, a
    k.act_low, k.act_high = float(model.args.action_low), float(model.args.action_high)
    k.means, k.eps, k.std, k.action = means.data_ptr(), eps.data_ptr(), cache[means.device].data_ptr(), out.data_ptr()
    if env_action is not None:
        k.env_action = env_action.data_ptr()
    _lib.check(_lib.load().flexnet_agent_sum_explore(C.byref(k), C.c_void_p(th.cuda.current_stream().cuda_stream)),
               "flexnet_agent_sum_explore")
    return out


class _SumBroadcastAgentsFn(th.autograd.Function):
    """y[b, i, :] = ((x[b, 0] + x[b, 1]) + ...) for every agent i — matd3.py:94-97's agent-summed mean handed back to every
    agent (nets.expand_agents of _sum_agents) — and its gradient, which is the same operation on the incoming gradient; one
    launch each way instead of 2 (n - 1) pointwise adds."""

    @staticmethod
    def forward(ctx, x, model):
        ctx.model = model
        return _sum_broadcast(model, x)

    @staticmethod
    def backward(ctx, g):
        return _sum_broadcast(ctx.model, g.contiguous()), None


def _sum_broadcast(model, x):
    import ctypes as C
    from . import _lib
    b, n, a = x.shape
    x = x.contiguous()
    out = th.empty(b, n, a, dtype=th.float32, device=x.device)
    k = _lib.FlexAgentSumArgs()
    k.n_envs, k.n_agents, k.act_dim = b, n, a
    k.means, k.action = x.data_ptr(), out.data_ptr()
    _lib.check(_lib.load().flexnet_agent_sum_explore(C.byref(k), C.c_void_p(th.cuda.current_stream().cuda_stream)),
               "flexnet_agent_sum_explore")
    return out


def _summed_mean_applies(model, means, status, exploration, actions_avail):
    """Training mode without exploration (the losses' policy evaluations): the summed mean to every agent, one launch."""
    return (status == "train" and not exploration and means.is_cuda and means.dtype == th.float32 and means.dim() == 3
            and means.size(-1) > 1 and getattr(actions_avail, "_flex_const", None) == 1.0
            and getattr(model, "fused_inference", False))


def _summed_exploration_applies(model, means, status, exploration, actions_avail, need_log_prob):
    """The one-launch form covers exactly: training-mode exploration with the bound enforced, every action available (the
    constant mask of the vectorised paths), no gradient through the means, nobody asking for the log-probability."""
    return (not need_log_prob and status == "train" and exploration and bool(model.args.action_enforcebound)
            and means.is_cuda and means.dtype == th.float32 and means.dim() == 3 and means.size(-1) > 1
            and not means.requires_grad and getattr(actions_avail, "_flex_const", None) == 1.0
            and getattr(model, "fused_inference", False))


def _sum_agents(x):
    """x.sum(dim=1, keepdim=True) over the (small) agent axis as n - 1 pointwise adds: the same numbers up to fp32
    summation order, and no ATen reduce_kernel in a captured rollout graph (util.GRAPH_DENYLIST is strict about those)."""
    parts = x.unbind(1)
    out = parts[0]
    for p in parts[1:]:
        out = out + p
    return out.unsqueeze(1)


class MATD3(MADDPG):
    """madrl/models/matd3.py:8-149: twin centralised critics realised as ONE shared network with
    a trailing 0/1 input flag (matd3.py:64-67), clipped-double-Q target min(Q1', Q2') (matd3.py:139-140) and a
    value loss averaged over the twins (matd3.py:148).  Bug-compatible with the reference's action selection, which
    sums the policy means over the AGENT axis before sampling (matd3.py:92-97)."""
    bootstrap_cacheable = False      # own get_loss (min of twins, target-smoothing noise drawn inside): values are per sub-update

    # since round 2 the GPU path of both losses reduces only through this project's fixed-order kernels (twin critic
    # nodes, flexnet_td_loss, flexnet_scaled_sum, pointwise agent sums): sub-updates replay as HIP graphs like MADDPG's
    graph_safe_updates = True

    def construct_value_net(self):
        """matd3.py:18-27: the MADDPG critic input plus the twin flag."""
        input_shape = (self.obs_dim + self.act_dim) * self.n_ + 1 + (self.n_ if self.args.agent_id else 0)
        count = 1 if self.args.shared_params else self.n_
        self.value_dicts = nn.ModuleList([MLPCritic(input_shape, 1, self.args) for _ in range(count)])

    def value(self, obs, act, critic_frozen=False, first_head_only=False):
        """matd3.py:33-86: returns cat([Q1, Q2], dim=0) of shape [2b, n, 1] (``first_head_only``: Q1 alone, [b, n, 1] — all
        the policy loss reads, matd3.py:121).  Same column-block evaluation as MADDPG.value; the twin flag only adds
        fc1.weight's last column to the second head's pre-activation."""
        if not self.args.shared_params:
            raise NotImplementedError("MATD3 is built for shared_params (default.yaml:26)")
        b, n, o, a = obs.size(0), self.n_, self.obs_dim, self.act_dim
        net = self.value_dicts[0]
        W, bias = net.fc1.weight, net.fc1.bias
        obs_cols = obs.reshape(b, n * o)
        if self.args.agent_id and obs.is_cuda:
            # GPU paths as single autograd nodes / fused tails (nets.py), every reduction in this project's kernels
            if (critic_frozen and first_head_only and act.requires_grad and th.is_grad_enabled()
                    and critic_policy_supported(net, obs_cols, act, n)):
                return CriticTail.apply_policy(obs_cols, act, net).view(b, n, 1)
            act_cols = act.detach().reshape(b, n * a)
            if (not act.requires_grad and th.is_grad_enabled() and W.requires_grad
                    and critic_replayed_supported(net, obs_cols, act_cols, n)):
                if W.shape[1] == n * o + n + n * a + 1 and self.fused_twin:
                    # both heads as one node: fc1's output formed once, its weight gradient taken once (nets._CriticReplayedTwinFn)
                    return CriticTail.apply_replayed_twin(obs_cols, act_cols, n, net).view(2 * b, n, 1)
                q1 = CriticTail.apply_replayed(obs_cols, act_cols, n, net)
                q2 = CriticTail.apply_replayed(obs_cols, act_cols, n, net, twin=True)
                return th.cat([q1.view(b, n, 1), q2.view(b, n, 1)], dim=0)
            if not th.is_grad_enabled() and critic_tail_supported(net, obs_cols.new_empty(1, self.hid_dim)):
                off = n * o
                shared = th.addmm(bias, obs_cols, W[:, :off].t())
                shared.addmm_(act_cols, W[:, off + n:off + n + n * a].t())
                ids = W[:, off:off + n].t()
                q1 = CriticTail.apply_composed(shared, ids, net)
                q2 = CriticTail.apply_composed(shared, ids + W[:, -1], net)
                return th.cat([q1.view(b, n, 1), q2.view(b, n, 1)], dim=0)
        act_det = act.detach()
        own = act - act_det
        off = n * o
        h = wide_batch_linear(obs_cols, W[:, :off]) + bias
        h = h.unsqueeze(1).expand(b, n, -1)
        if self.args.agent_id:
            h = h + W[:, off:off + n].t().unsqueeze(0)
            off += n
        W_act = W[:, off:off + n * a]
        h = h + (act_det.reshape(b, n * a) @ W_act.t()).unsqueeze(1)
        if act.requires_grad:
            h = h + th.einsum("bia,hia->bih", own, W_act.view(-1, n, a))
        flag = W[:, off + n * a]                                            # column of the 0/1 twin flag
        v1, _ = net.forward_from_hidden(h.reshape(b * n, -1), need_hidden=False)
        if first_head_only:
            return v1.view(b, n, 1)
        v2, _ = net.forward_from_hidden((h + flag).reshape(b * n, -1), need_hidden=False)
        return th.cat([v1.view(b, n, 1), v2.view(b, n, 1)], dim=0)

    def get_actions(self, obs, status, exploration, actions_avail, target=False, last_hid=None, clip=False, need_log_prob=True):
        """matd3.py:88-111 (continuous branch).  ``need_log_prob=False`` (the losses and the vectorised rollout never read
        it): exploration in one launch where ``_summed_exploration_applies``."""
        pol = self.target_net.policy if (target and self.args.target) else self.policy
        means, log_stds, hiddens = pol(obs, last_hid=last_hid)
        if _summed_exploration_applies(self, means, status, exploration, actions_avail, need_log_prob):
            restore_actions = summed_exploration(self, means)
            return restore_actions[:, :1], restore_actions, None, (means, log_stds), hiddens
        if _summed_mean_applies(self, means, status, exploration, actions_avail):
            restore_actions = _SumBroadcastAgentsFn.apply(means, self)
            return restore_actions[:, :1], restore_actions, None, (means, log_stds), hiddens
        avail = actions_avail.to(means.device)
        means = means.masked_fill(avail == 0, 0.0)
        log_stds = log_stds.masked_fill(avail == 0, 0.0)
        if means.size(-1) > 1:                                              # matd3.py:94-96: sum over dim=1 (agents)
            means_, log_stds_ = _sum_agents(means), _sum_agents(log_stds)
        else:
            means_, log_stds_ = means, log_stds
        actions, log_prob_a = select_action(self.args, means_, status=status, exploration=exploration,
                                            info={"clip": clip, "log_std": log_stds_})
        if getattr(actions_avail, "_flex_const", None) == 1.0 and actions.size(1) == 1:
            # every action available (env:721-730): the mask is 1; the agent-summed action goes to every agent
            restore_actions = expand_agents(actions, self.n_)
        else:
            restore_actions = (1.0 - (avail == 0).float()) * actions
        return actions, restore_actions, log_prob_a, (means, log_stds), hiddens

    def get_loss(self, batch, need="both"):
        """matd3.py:113-149.  ``need`` = "value" / "policy" (what a sub-update asks for) on the GPU: the reward BatchNorm
        and both TD terms through flexnet_td_loss (batch statistics once, running statistics moved once), the policy loss
        through the first head's action-gradient node; "both" is the reference's call."""
        on_gpu = (isinstance(batch.reward, th.Tensor) and batch.reward.is_cuda and self.fused_inference
                  and th.is_grad_enabled() and need in ("value", "policy") and self.args.reward_normalisation
                  and batchnorm_stats_supported(self.batchnorm, batch.reward))
        state, actions, _, _, _, rewards, next_state, done, _, actions_avail, last_hids, hids = \
            self.unpack_data(batch, normalise_reward=not on_gpu)
        b = state.size(0)
        policy_loss = value_loss = action_out = None
        if need in ("both", "policy"):
            if on_gpu:                      # the policy loss never reads the reward: the module's bookkeeping only
                batchnorm_update_running_stats(self.batchnorm, rewards)
            _, actions_pol, _, action_out, _ = self.get_actions(state, status="train", exploration=False,
                                                                actions_avail=actions_avail, target=False,
                                                                last_hid=last_hids)
            if on_gpu:
                advantages = self.value(state, actions_pol, critic_frozen=True, first_head_only=True).reshape(-1, self.n_)
            else:
                advantages = self.value(state, actions_pol)[:b].reshape(-1, self.n_)      # first head only
            if self.args.normalize_advantages:
                advantages = self.batchnorm(advantages)
            policy_loss = mean_all(advantages, sign=-1.0)
        if need in ("both", "value"):
            with th.no_grad():
                _, next_actions, _, _, _ = self.get_actions(next_state, status="train", exploration=True,
                                                            actions_avail=actions_avail, target=not self.args.double_q,
                                                            last_hid=hids, clip=True, need_log_prob=False)
                nxt = self.target_net.value(next_state, next_actions)
                next_min = th.min(nxt[:b].reshape(-1, self.n_), nxt[b:].reshape(-1, self.n_))
            cur = self.value(state, actions)
            values1, values2 = cur[:b].reshape(-1, self.n_), cur[b:].reshape(-1, self.n_)
            if on_gpu and td_loss_supported(values1, next_min, rewards, done, self.batchnorm):
                # matd3.py:141-148: both twins against the same normalised return; the second term must not move the
                # BatchNorm's running statistics again
                value_loss = 0.5 * (td_loss(values1, next_min, rewards, done, self.args.gamma, self.batchnorm)
                                    + td_loss(values2, next_min, rewards, done, self.args.gamma, self.batchnorm,
                                              update_stats=False))
            else:
                if on_gpu:                  # the raw reward was handed on: normalise it here
                    with th.no_grad():
                        rewards = sync_batchnorm(self.batchnorm, rewards.contiguous())
                returns = rewards + self.args.gamma * (1 - done) * next_min
                assert returns.size() == values1.size() == values2.size()
                value_loss = 0.5 * (mean_all((returns - values1).pow(2)) + mean_all((returns - values2).pow(2)))
        return policy_loss, value_loss, action_out


MATD3.fused_twin = True             # (tests switch it off to compare with the two single-head nodes)


class IDDPG(MADDPG):
    """madrl/models/iddpg.py:7-83 with the loss of madrl/learning_algorithms/ddpg.py:14-37:
    independent critics Q_i(o_i, a_i) on the agent's own observation and action (plus its one-hot id), the same
    DDPG losses as MADDPG, and the agent-summed action selection it shares with MATD3 (iddpg.py:66-71)."""

    graph_safe_updates = True       # as MATD3: no PyTorch reduction is left on the GPU path of either loss

    def construct_value_net(self):
        """iddpg.py:17-26"""
        input_shape = self.obs_dim + self.act_dim + (self.n_ if self.args.agent_id else 0)
        count = 1 if self.args.shared_params else self.n_
        self.value_dicts = nn.ModuleList([MLPCritic(input_shape, 1, self.args) for _ in range(count)])

    def value(self, obs, act, critic_frozen=False):
        """iddpg.py:32-59: rows [o_i | onehot(i) | a_i] -> [b, n, 1]."""
        b = obs.size(0)
        if self.args.agent_id:
            ids = th.eye(self.n_, device=obs.device, dtype=obs.dtype).expand(b, -1, -1)
            obs = th.cat((obs, ids), dim=-1)
        inputs = th.cat((obs, act), dim=-1)
        if self.args.shared_params:
            net = self.value_dicts[0]
            rows = inputs.reshape(b * self.n_, -1)
            if rows.is_cuda and rows.shape[0] >= WGRAD_MIN_ROWS:
                # update batches: first layer with the batch-reduced weight gradient of csrc/wgrad.hip (the library's
                # dW = dY^T X over 163 840 rows took 0.8 ms), the rest of the critic in the fused tail kernels
                v, _ = net.forward_from_hidden(tall_linear(rows, net.fc1.weight, net.fc1.bias), need_hidden=False)
            else:
                v, _ = net(rows, None)
            return v.view(b, self.n_, -1)
        return th.stack([net(inputs[:, i, :], None)[0] for i, net in enumerate(self.value_dicts)], dim=1)

    def get_actions(self, state, status, exploration, actions_avail, target=False, last_hid=None, need_log_prob=True):
        """iddpg.py:61-83 (continuous branch): the means are summed over the agent axis before sampling."""
        pol = self.target_net.policy if (target and self.args.target) else self.policy
        means, log_stds, hiddens = pol(state, last_hid=last_hid)
        if _summed_exploration_applies(self, means, status, exploration, actions_avail, need_log_prob):
            restore_actions = summed_exploration(self, means)
            return restore_actions[:, :1], restore_actions, None, (means, log_stds), hiddens
        if _summed_mean_applies(self, means, status, exploration, actions_avail):
            restore_actions = _SumBroadcastAgentsFn.apply(means, self)
            return restore_actions[:, :1], restore_actions, None, (means, log_stds), hiddens
        if means.size(-1) > 1:
            means_, log_stds_ = _sum_agents(means), _sum_agents(log_stds)
        else:
            means_, log_stds_ = means, log_stds
        actions, log_prob_a = select_action(self.args, means_, status=status, exploration=exploration,
                                            info={"log_std": log_stds_})
        if getattr(actions_avail, "_flex_const", None) == 1.0 and actions.size(1) == 1:
            restore_actions = expand_agents(actions, self.n_)       # every action available: the mask is 1
        else:
            restore_actions = (1.0 - (actions_avail.to(means.device) == 0).float()) * actions
        return actions, restore_actions, log_prob_a, (means, log_stds), hiddens


class SAFEMADDPG(MADDPG):
    """safemaddpg.py:14-299: MADDPG whose get_actions passes the proposed action through the safety layer.

    The reference solves a Pyomo QP with Gurobi per call (safemaddpg.py:176-299) using a pickled sklearn
    regressor (safemaddpg.py:27); here the layer is the HIP closed-form projection
    (``flexenv_safety_project``) and the regressor's row sums come from ``safety_signal.fit_voltage_predictor``
    (or any (s_p, s_q, beta) triple the caller passes as ``predictor``)."""

    def __init__(self, args, env, target_net=None, predictor=None):
        super().__init__(args, target_net)
        self.env = env
        self.V_min = args.v_min
        self.V_max = args.v_max
        self.solver_interventions = 0
        self.solver_infeasible = 0
        if predictor is None:
            from .safety_signal import fit_voltage_predictor
            vec = env.vec if hasattr(env, "vec") else env
            predictor = fit_voltage_predictor(vec.net, device=vec.device).building_terms(vec.net)
        self.predictor = predictor          # (s_p [n], s_q [n], beta [n]) per building

    def safety_layer_optimization(self, proposed_actions):
        """safemaddpg.py:176-299 on every env of the batch: returns [B, 4n] type-major adjusted actions."""
        vec = self.env.vec if hasattr(self.env, "vec") else self.env
        s_p, s_q, beta = self.predictor
        adjusted, hit = vec.safety_project(proposed_actions.detach().reshape(-1, self.n_, self.act_dim),
                                           s_p, s_q, beta, self.V_min, self.V_max)
        self._last_intervened = hit
        return adjusted

    def get_actions(self, state, status, exploration, actions_avail, target=False, last_hid=None):
        """safemaddpg.py:90-111.  During a loss evaluation (batch != env batch) the reference solves the QP on
        batch element 0 against the live env and discards the result (safemaddpg.py:118-121);
        that call is skipped here — it has no observable effect."""
        actions, restore_actions, log_prob_a, action_out, hiddens = super().get_actions(
            state, status, exploration, actions_avail, target, last_hid)
        vec = self.env.vec if hasattr(self.env, "vec") else self.env
        if state.size(0) != vec.n_envs:
            return actions, restore_actions, log_prob_a, action_out, hiddens
        adjusted = self.safety_layer_optimization(restore_actions).to(th.float32)   # safemaddpg.py:106-109
        return adjusted, restore_actions, log_prob_a, action_out, hiddens

    # NOT the reference's behaviour (off by default; no parity claim is made with it on).  The reference hands the safety
    # layer's type-major physical-unit vector (safemaddpg.py:297) through translate_action (util.py:125-128: clamp to [0, 1],
    # then 0.5 (a + 1)) to an env that re-reads it agent-major as raw values (env:268-274) — SURVEY A13.  Every entry then
    # arrives >= 0.5: the load reduction is clipped to its maximum, charging and discharging cancel at their maxima, and q
    # lands in [0.5, 0.5 + c] pu, so the policy cannot move the environment (tools/learning_curve.py: the test reward of
    # SAFEMADDPG stays within 3e-4 of -0.097 from the untrained policy to episode 400, at either batch size).  With
    # ``intended_actions`` the adjusted (percentage, charge, discharge, q) of building i reaches the env as is, agent-major:
    # what safemaddpg.py:90-111 evidently means to do.
    intended_actions = False

    def env_action(self, action):
        if self.intended_actions:
            return action.detach().reshape(-1, self.act_dim, self.n_).transpose(1, 2).to(th.float32).contiguous()
        # the type-major flat vector is re-read agent-major by env.step (safemaddpg.py:297 vs env:268-274, A13)
        return scale_action(self.args, action.detach()).reshape(-1, self.n_, self.act_dim)
