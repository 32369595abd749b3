"""Host side of the flexibility-provision environment above the C ABI.

``VecFlexProvisionEnv`` runs N independent environments on one MI355X (one
wavefront each); ``FlexibilityProvisionEnv`` is the N=1 view with exactly the
surface of madrl/environments/flex_provision/flexibility_provision_env.py so it
drops into train_agent.py / run_env.py (SURVEY.md §8b lists the call sites).

All arithmetic happens in the HIP library; this file only owns device buffers
(torch tensors), draws the host RNG stream the reference draws (N=1 view) and
converts to the reference's return types.
"""
from __future__ import annotations

import ctypes as C
from collections import namedtuple
from math import acos, tan

import numpy as np
import torch

from . import _lib
from .network import build_tables, create_network
from .series import SeriesTable, make_synthetic_series

DEFAULT_ENV_ARGS = dict(  # madrl/args/env_args/flex_provision.yaml:3-33
    history=24, pv_scale=0.15, demand_scale=1.0, reactive_scale=1.0, v_max=1.1, v_min=0.9, data_path="./data",
    episode_limit=96, action_low=0, action_high=1.0, seed=0, e_min=0.0, e_max=0.025, pv_cost=0.05, ess_cost=0.03,
    discomfort_coeff=0.15, voltage_coeff=1.0, p_ch_max=0.005, p_dis_max=0.005, eta_ch=0.9, eta_dis=0.9,
    cos_phi_max=0.95, max_power_reduction=0.5, sample_interval="15min", buildings=[5, 10, 15, 20, 25],
    pv_nodes=[5, 10, 15, 20, 25], ess_nodes=[5, 10, 15, 20, 25], v_nom=12.66, s_nom=1000, pv_cap=0.15,
)


def convert(dictionary):
    """env:14-15"""
    return namedtuple("GenericDict", dictionary.keys())(**dictionary)


class ActionSpace(object):  # env:17-20
    def __init__(self, low, high):
        self.low = low
        self.high = high


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class VecFlexProvisionEnv:
    """N environments, device-resident.  Tensors in, tensors out, nothing syncs the host.

    reset(mask=None, spec=None) -> obs [N, n_agents, 6*history] float32
    step(actions [N, n_agents, 4]) -> (reward [N] f64, done [N] uint8, info [N, 7] f64)
    get_obs() -> obs (stateful like env:370-403)
    """

    def __init__(self, env_args=None, n_envs=1, device="cuda:0", net=None, series=None, pf_tol=1e-12,
                 pf_max_iter=20, warm_start=False, solver=_lib.FLEX_SOLVER_SWEEP, seed=None, sweep_accel=True):
        args = dict(DEFAULT_ENV_ARGS)
        args.update(env_args or {})
        self.args_dict = args
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.FlexLibraryError("VecFlexProvisionEnv needs a GPU: the product path has no CPU fallback")
        self.device = torch.device(device)
        self.n_envs = int(n_envs)
        self.net = net if net is not None else create_network(args)
        self.tables = build_tables(self.net)
        self.n_bus = self.tables.n_bus
        self.n_agents = len(self.net["buildings"])
        self.n_actions = 4
        self.history = int(args["history"])
        self.episode_limit = int(args["episode_limit"])
        if series is None:
            series = make_synthetic_series(self.net, pv_scale=args["pv_scale"])
        if not isinstance(series, SeriesTable):
            raise TypeError("series must be a SeriesTable")
        if series.n_bus != self.n_bus or series.n_agents != self.n_agents:
            raise ValueError("series table does not match the network")
        self.series = series
        with torch.cuda.device(self.device):
            self.series_dev = torch.from_numpy(series.table).to(self.device)
        cfg = _lib.FlexCfg()
        cfg.n_agents = self.n_agents
        cfg.history = self.history
        cfg.episode_limit = self.episode_limit
        cfg.per_hour = series.per_hour
        cfg.n_start_days = series.n_start_days(self.episode_limit)
        cfg.raw_actions = 1 if args.get("alg") == "safemaddpg" else 0      # env:43,268
        cfg.pf_max_iter = pf_max_iter
        cfg.solver = solver
        cfg.warm_start = 1 if warm_start else 0
        cfg.no_sweep_accel = 0 if sweep_accel else 1       # include/flexenv.h: plain sweeps, for A/B measurements and tests
        for k in ("v_min", "v_max", "e_min", "e_max", "p_ch_max", "p_dis_max", "eta_ch", "eta_dis",
                  "max_power_reduction", "pv_cost", "ess_cost", "discomfort_coeff", "voltage_coeff",
                  "action_low", "action_high"):
            setattr(cfg, k, float(args[k]))
        cfg.tan_phi = tan(acos(args["cos_phi_max"]))                        # env:623
        cfg.dt = 24 / self.episode_limit                                    # pf.py:23-24
        cfg.fail_penalty = 200.0                                            # env:336
        cfg.pf_tol = pf_tol
        cfg.seed = int(args["seed"] if seed is None else seed)
        self.cfg = cfg
        t = self.tables
        self._keep = dict(
            parent=np.ascontiguousarray(t.parent, np.int32), level=np.ascontiguousarray(t.level, np.int32),
            child=np.ascontiguousarray(t.child, np.int32), r=np.ascontiguousarray(t.r, np.float64),
            x=np.ascontiguousarray(t.x, np.float64), agent_bus=np.ascontiguousarray(t.agent_bus, np.int32))
        nf = _lib.NetFix()
        nf.n_bus, nf.slack, nf.n_levels, nf.max_children = t.n_bus, t.slack, t.n_levels, t.max_children
        nf.parent = self._keep["parent"].ctypes.data_as(C.POINTER(C.c_int32))
        nf.level = self._keep["level"].ctypes.data_as(C.POINTER(C.c_int32))
        nf.child = self._keep["child"].ctypes.data_as(C.POINTER(C.c_int32))
        nf.r = self._keep["r"].ctypes.data_as(C.POINTER(C.c_double))
        nf.x = self._keep["x"].ctypes.data_as(C.POINTER(C.c_double))
        nf.agent_bus = self._keep["agent_bus"].ctypes.data_as(C.POINTER(C.c_int32))
        self.netfix = nf
        st = _lib.SeriesTab(self.series_dev.data_ptr(), series.rows, series.cols)
        handle = C.c_void_p()
        _lib.check(self.lib.flexenv_create(C.byref(cfg), C.byref(nf), C.byref(st), self.n_envs,
                                           self.device.index or 0, C.byref(handle)), "flexenv_create")
        self.handle = handle
        self.obs_size = self.lib.flexenv_obs_size(handle)
        self.state_size = self.lib.flexenv_state_size(handle)
        N, na = self.n_envs, self.n_agents
        dev = self.device
        self.reward = torch.zeros(N, dtype=torch.float64, device=dev)
        self.done = torch.zeros(N, dtype=torch.uint8, device=dev)
        self.info = torch.zeros(N, _lib.FLEX_INFO_W, dtype=torch.float64, device=dev)
        self.failed = torch.zeros(N, dtype=torch.uint8, device=dev)
        self.obs = torch.zeros(N, na, self.obs_size, dtype=torch.float32, device=dev)
        self.calls = 0          # reset() / step() calls made through this object (a captured graph replays none)

    def close(self):
        if getattr(self, "handle", None):
            self.lib.flexenv_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- reset / step / obs ------------------------------------------------------------
    def _dtype_tag(self, t):
        if t.dtype == torch.float32:
            return _lib.FLEX_F32
        if t.dtype == torch.float64:
            return _lib.FLEX_F64
        raise TypeError(f"unsupported dtype {t.dtype}")

    def reset(self, mask=None, spec=None, obs_out=None, want_obs=True):
        """spec: dict with any of day/hour/interval (int32 [N]) and e0 [N,na], a0 [N,4na] (f64) device
        tensors; missing items come from the Philox reset stream."""
        self.calls += 1
        rs = None
        keep = []
        if spec:
            rs = _lib.ResetSpec()
            for k, dt in (("day", torch.int32), ("hour", torch.int32), ("interval", torch.int32),
                          ("e0", torch.float64), ("a0", torch.float64)):
                v = spec.get(k)
                if v is not None:
                    v = torch.as_tensor(v, dtype=dt, device=self.device).contiguous()
                    keep.append(v)
                    setattr(rs, k, v.data_ptr())
        if mask is not None:
            mask = torch.as_tensor(mask, device=self.device).to(torch.uint8).contiguous()
        out = None
        if want_obs:
            out = self.obs if obs_out is None else obs_out
        _lib.check(self.lib.flexenv_reset(self.handle, _ptr(mask), C.byref(rs) if rs is not None else None,
                                          _ptr(out), self._dtype_tag(out) if out is not None else 0,
                                          _ptr(self.failed), _stream()), "flexenv_reset")
        return out

    def step(self, actions, obs_out=None, fuse_obs=False, auto_reset=False, obs_ring=None, replay_sink=False, obs_rows=False):
        """auto_reset: environments that terminate in this step restart inside the same launch (their row of the
        fused observation is then the first observation of the new episode).
        fuse_obs: the get_obs() that follows every step (model.py:223) in the same launch, as a stacked copy into
        ``self.obs`` / ``obs_out``.  obs_rows: the same get_obs() as a ROW PUSH — the step appends its 6-feature row per agent
        to the environment's history (120 B per env instead of a 2 880 B copy) and consumers read the stacked observation in
        place (``obs_source()``: the policy kernels do) or materialise it on demand (``obs_view()``).  obs_ring: base
        tensor of the ROW ring registered with set_obs_ring — implies obs_rows, and the step also files one record per
        (env, agent) in the slab after the cursor (include/flexenv.h: FLEX_STEP_OBS_RING)."""
        self.calls += 1
        if actions.device != self.device:
            actions = actions.to(self.device)
        actions = actions.contiguous()
        if actions.numel() != self.n_envs * self.n_agents * 4:
            raise ValueError(f"actions must have {self.n_envs}x{self.n_agents}x4 elements, got {tuple(actions.shape)}")
        out = None
        flags = _lib.FLEX_STEP_AUTORESET if auto_reset else 0
        if obs_ring is not None:
            out, flags = obs_ring, flags | _lib.FLEX_STEP_OBS_RING
            if replay_sink:                  # the step files its own transition (set_replay_sink)
                flags |= _lib.FLEX_STEP_REPLAY_SINK
        elif obs_rows:
            flags |= _lib.FLEX_STEP_OBS_ROWS
        elif fuse_obs:
            out = self.obs if obs_out is None else obs_out
        _lib.check(self.lib.flexenv_step(self.handle, _ptr(actions), self._dtype_tag(actions), _ptr(self.reward),
                                         _ptr(self.done), _ptr(self.info), _ptr(self.failed), _ptr(out),
                                         self._dtype_tag(out) if out is not None else 0, flags, _stream()), "flexenv_step")
        return self.reward, self.done, self.info

    def step_many(self, actions, steps=None, auto_reset=False, want_info=True, out=None, carry=True):
        """``steps`` consecutive vector steps on a GIVEN action sequence in ONE launch (include/flexenv.h:
        flexenv_step_many) — the reference's open-loop episode runner, run_env.py:78-92 (sampled actions, step(), per-step
        records), for every environment at once.  ``actions``: [P, N, n_agents, 4]; step k uses slab k mod P (``steps``
        defaults to P).  get_obs() is the row push of ``step(..., obs_rows=True)``; state, history and outputs equal those of
        ``steps`` such calls bit for bit.  Returns (reward [steps, N], done [steps, N], info [steps, N, 7] or None,
        failed [steps, N]) — preallocated tensors may be handed in as ``out`` (same tuple; HIP-graph capture)."""
        if actions.device != self.device:
            actions = actions.to(self.device)
        actions = actions.contiguous()
        per = self.n_envs * self.n_agents * 4
        if actions.numel() == 0 or actions.numel() % per:
            raise ValueError(f"actions must be [P, {self.n_envs}, {self.n_agents}, 4], got {tuple(actions.shape)}")
        period = actions.numel() // per
        steps = period if steps is None else int(steps)
        if steps < 1:
            raise ValueError("steps must be >= 1")
        if out is None:
            reward = torch.empty(steps, self.n_envs, dtype=torch.float64, device=self.device)
            done = torch.empty(steps, self.n_envs, dtype=torch.uint8, device=self.device)
            info = torch.empty(steps, self.n_envs, _lib.FLEX_INFO_W, dtype=torch.float64, device=self.device) if want_info else None
            failed = torch.empty(steps, self.n_envs, dtype=torch.uint8, device=self.device)
        else:
            reward, done, info, failed = out
            for t, w, dt in ((reward, 1, torch.float64), (done, 1, torch.uint8), (info, _lib.FLEX_INFO_W, torch.float64),
                             (failed, 1, torch.uint8)):
                if t is not None and not (t.device == self.device and t.is_contiguous() and t.dtype == dt
                                          and t.numel() == steps * self.n_envs * w):
                    raise ValueError("step_many: `out` tensors must be contiguous device tensors of [steps, N(, 7)]")
        self.calls += steps
        flags = _lib.FLEX_STEP_OBS_ROWS | (_lib.FLEX_STEP_AUTORESET if auto_reset else 0) | (0 if carry else _lib.FLEX_STEP_MANY_NO_CARRY)
        _lib.check(self.lib.flexenv_step_many(self.handle, _ptr(actions), self._dtype_tag(actions), period, steps, _ptr(reward),
                                              _ptr(done), _ptr(info), _ptr(failed), flags, _stream()), "flexenv_step_many")
        return reward, done, info, failed

    def step_many_prepared(self, actions, steps=None, auto_reset=False, want_info=True, out=None, stream=None):
        """``step_many`` with its arguments checked and marshalled ONCE: returns ``(launch, (reward, done, info, failed))`` where
        ``launch()`` issues the same flexenv_step_many call again on ``stream`` (default: the stream current now) — for a caller
        that replays one sequence length on fixed buffers, where the ~10 us of per-call checks would sit in front of a launch of
        a few dozen steps.  The tensors are kept alive by the closure."""
        res = self.step_many(actions, steps=steps, auto_reset=auto_reset, want_info=want_info, out=out)   # checked, run once
        reward, done, info, failed = res
        actions = actions.to(self.device).contiguous()
        per = self.n_envs * self.n_agents * 4
        period = actions.numel() // per
        n = int(reward.shape[0])
        flags = _lib.FLEX_STEP_OBS_ROWS | (_lib.FLEX_STEP_AUTORESET if auto_reset else 0)
        st = C.c_void_p((stream or torch.cuda.current_stream()).cuda_stream)
        args = (self.handle, _ptr(actions), self._dtype_tag(actions), period, n, _ptr(reward), _ptr(done), _ptr(info), _ptr(failed),
                flags, st)
        fn = self.lib.flexenv_step_many
        keep = (actions, reward, done, info, failed)

        def launch(_fn=fn, _args=args, _keep=keep):
            rc = _fn(*_args)
            if rc:
                _lib.check(rc, "flexenv_step_many")
            self.calls += n

        return launch, res

    def rollout_burst(self, actor_args, steps, obs_ring, safety=None):
        """``steps`` vector steps of policy + environment in ONE launch (include/flexenv.h: flexenv_rollout_burst):
        ``actor_args`` is the FlexActorArgs of the ring-mode policy call the burst replaces (nets.fused_actor_forward builds
        it), the env's obs ring and replay sink are configured as for ``step(..., obs_ring=..., replay_sink=True)``.
        ``safety`` (SAFEMADDPG): dict(s_p, s_q, beta: float64 device tensors [n_agents]; v_min, v_max, penalty; adjusted
        [N, 4 n] float64 and env_action [N, 4 n] float32 device tensors; act_low, act_high) — safety_project(...,
        env_action_range=...) between policy and step, inside the launch."""
        self.calls += int(steps)
        sf = None
        if safety is not None:
            sf = _lib.FlexBurstSafety()
            for k in ("s_p", "s_q", "beta", "adjusted", "env_action"):
                t = safety[k]
                want = torch.float32 if k == "env_action" else torch.float64
                if not (t.is_cuda and t.is_contiguous() and t.dtype == want):
                    raise ValueError(f"rollout_burst safety: {k} must be a contiguous {want} device tensor")
                setattr(sf, k, t.data_ptr())
            if safety["adjusted"].numel() != self.n_envs * 4 * self.n_agents or safety["env_action"].numel() != self.n_envs * 4 * self.n_agents:
                raise ValueError("rollout_burst safety: adjusted / env_action must hold n_envs x 4 n_agents elements")
            sf.v_min, sf.v_max, sf.penalty = float(safety["v_min"]), float(safety["v_max"]), float(safety.get("penalty", 1000.0))
            sf.act_low, sf.act_high = float(safety["act_low"]), float(safety["act_high"])
        _lib.check(self.lib.flexenv_rollout_burst(self.handle, C.byref(actor_args), _ptr(self.reward), _ptr(self.done),
                                                  _ptr(self.info), _ptr(self.failed), _ptr(obs_ring), int(steps),
                                                  C.byref(sf) if sf is not None else None, _stream()),
                   "flexenv_rollout_burst")

    def set_step_counter(self, counter, modulo=0):
        """Every later step() adds 1 to ``counter[0]`` (int64 device tensor, or None to switch it off) from inside the
        step kernel, wrapping to 0 at ``modulo`` (0 = never) — how a replayed HIP graph keeps the replay ring's cursor
        moving without a launch of its own (include/flexenv.h: flexenv_set_step_counter).  The tensor is kept alive for as
        long as the env points at it."""
        if counter is not None and not (counter.is_cuda and counter.dtype == torch.int64 and counter.is_contiguous()):
            raise ValueError("step counter must be a contiguous int64 device tensor")
        _lib.check(self.lib.flexenv_set_step_counter(self.handle, _ptr(counter), int(modulo)), "flexenv_set_step_counter")
        self._step_counter = counter

    def set_obs_ring(self, cursor, slab_stride, slabs):
        """Register a slab ring for ``step(..., obs_ring=base)``: that launch writes its observations into slab
        ``(cursor[0] + 1) % slabs`` of the ring starting at ``base`` (slabs ``slab_stride`` floats apart), read on the
        device (include/flexenv.h: flexenv_set_obs_ring).  ``slabs = 0`` switches it off."""
        if slabs and not (cursor.is_cuda and cursor.dtype == torch.int64 and cursor.is_contiguous()):
            raise ValueError("ring cursor must be a contiguous int64 device tensor")
        _lib.check(self.lib.flexenv_set_obs_ring(self.handle, _ptr(cursor) if slabs else None, int(slab_stride), int(slabs)),
                   "flexenv_set_obs_ring")
        self._obs_cursor = cursor if slabs else None

    def set_replay_sink(self, policy_action, hid_new, small_ring, hid_ring, acc, cursor_out=None, aux_counter=None):
        """Register where ``step(..., obs_ring=..., replay_sink=True)`` files a step's transition (include/flexenv.h:
        FlexReplaySink): the policy's action and new recurrent state are read from ``policy_action`` [N, act_w] /
        ``hid_new`` [N, hid_w]; the small record goes to slab cursor of ``small_ring``, the masked recurrent state to slab
        cursor + 1 of ``hid_ring``, the episode statistics to the per-environment running sums ``acc`` [N, 10] (fp64).
        ``None`` for the first argument switches the sink off.  The tensors are kept alive with the env."""
        if policy_action is None:
            _lib.check(self.lib.flexenv_set_replay_sink(self.handle, None), "flexenv_set_replay_sink")
            self._sink_keep = None
            return
        sk = _lib.FlexReplaySink()
        N = self.n_envs
        pa, hn = policy_action.view(N, -1), hid_new.view(N, -1)
        for t in (pa, hn, small_ring, hid_ring, acc):
            if not (t.is_cuda and t.is_contiguous()):
                raise ValueError("replay sink tensors must be contiguous device tensors")
        if acc.dtype != torch.float64 or acc.shape != (N, 10):
            raise ValueError("acc must be a float64 [n_envs, 10] tensor")
        sk.policy_action, sk.hid_new, sk.small_ring, sk.hid_ring, sk.acc = (pa.data_ptr(), hn.data_ptr(), small_ring.data_ptr(),
                                                                             hid_ring.data_ptr(), acc.data_ptr())
        sk.cursor_out = _ptr(cursor_out)
        sk.aux_counter = _ptr(aux_counter)
        sk.act_w, sk.hid_w, sk.small_w = pa.shape[1], hn.shape[1], small_ring.shape[-1]
        _lib.check(self.lib.flexenv_set_replay_sink(self.handle, C.byref(sk)), "flexenv_set_replay_sink")
        self._sink_keep = (policy_action, hid_new, small_ring, hid_ring, acc, cursor_out, aux_counter)

    def get_obs(self, obs_out=None):
        out = self.obs if obs_out is None else obs_out
        _lib.check(self.lib.flexenv_obs(self.handle, _ptr(out), self._dtype_tag(out), _stream()), "flexenv_obs")
        return out

    def obs_view(self, obs_out=None):
        """The stacked observation as the last push left it (env:387-401), WITHOUT appending to the history: what
        ``step(..., obs_rows=True)`` leaves for consumers that want the [N, n_agents, 6 * history] copy."""
        out = self.obs if obs_out is None else obs_out
        _lib.check(self.lib.flexenv_obs_view(self.handle, _ptr(out), self._dtype_tag(out), _stream()), "flexenv_obs_view")
        return out

    def obs_source(self):
        """Where the observation history lives on the device (include/flexenv.h: FlexObsSource), for kernels that read the
        stacked observation in place (nets.fused_actor_forward(..., obs_source=...))."""
        src = _lib.FlexObsSource()
        _lib.check(self.lib.flexenv_obs_source(self.handle, C.byref(src)), "flexenv_obs_source")
        return src

    def get_state(self):
        out = torch.empty(self.n_envs, self.state_size, dtype=torch.float64, device=self.device)
        _lib.check(self.lib.flexenv_state(self.handle, _ptr(out), _stream()), "flexenv_state")
        return out

    def peek(self, name):
        f = _lib.PEEK[name]
        N, na = self.n_envs, self.n_agents
        if f == 0:
            out = torch.empty(N, self.n_bus, dtype=torch.float64, device=self.device)
        elif f <= 7:
            out = torch.empty(N, na, dtype=torch.float64, device=self.device)
        elif f == 8:
            out = torch.empty(N, dtype=torch.float64, device=self.device)
        else:
            out = torch.empty(N, dtype=torch.int32, device=self.device)
        _lib.check(self.lib.flexenv_peek(self.handle, f, _ptr(out), _stream()), "flexenv_peek")
        return out

    def poke(self, name, value):
        v = torch.as_tensor(value, dtype=torch.float64, device=self.device).contiguous()
        _lib.check(self.lib.flexenv_poke(self.handle, _lib.PEEK[name], _ptr(v), _stream()), "flexenv_poke")

    def safety_project(self, proposed, s_p, s_q, beta, v_min, v_max, penalty=1000.0, env_action_range=None, want_hit=True):
        """SAFEMADDPG.safety_layer_optimization on every env (safemaddpg.py:176-299).  Returns
        (adjusted [N, 4*n_agents] f64 type-major, intervened [N] uint8).
        ``env_action_range`` = (low, high): a third result, [N, 4*n_agents] f32 — translate_action (util.py:125-128) of the
        fp32 cast of ``adjusted``, what env.step is fed next, from the same launch (include/flexenv.h:
        flexenv_safety_project_env).  ``want_hit=False``: no intervention flags (their memset is a launch of its own)."""
        proposed = proposed.contiguous()
        adj = torch.empty(self.n_envs, 4 * self.n_agents, dtype=torch.float64, device=self.device)
        hit = torch.empty(self.n_envs, dtype=torch.uint8, device=self.device) if want_hit else None
        sp, sq, bt = (torch.as_tensor(x, dtype=torch.float64, device=self.device).contiguous() for x in (s_p, s_q, beta))
        if env_action_range is None:
            _lib.check(self.lib.flexenv_safety_project(self.handle, _ptr(proposed), self._dtype_tag(proposed), _ptr(sp),
                                                       _ptr(sq), _ptr(bt), float(v_min), float(v_max), float(penalty),
                                                       _ptr(adj), _ptr(hit), _stream()), "flexenv_safety_project")
            return adj, hit
        low, high = env_action_range
        env_action = torch.empty(self.n_envs, 4 * self.n_agents, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.flexenv_safety_project_env(self.handle, _ptr(proposed), self._dtype_tag(proposed), _ptr(sp),
                                                       _ptr(sq), _ptr(bt), float(v_min), float(v_max), float(penalty),
                                                       _ptr(adj), _ptr(hit), float(low), float(high), _ptr(env_action),
                                                       _stream()), "flexenv_safety_project_env")
        return adj, hit, env_action

    # size getters, env:708-738
    def get_obs_size(self):
        return self.obs_size

    def get_state_size(self):
        return self.state_size

    def get_total_actions(self):
        return self.n_actions

    def get_num_of_agents(self):
        return self.n_agents


def pf_solve_batch(net, pnet, qnet, tol=1e-12, max_iter=20, want_branch=False, solver=_lib.FLEX_SOLVER_SWEEP):
    """power_flow_solver_simplified (pf.py:115-192) on a batch of net loads [n, n_bus] (device f64)."""
    lib = _lib.load()
    t = build_tables(net)
    keep = dict(parent=np.ascontiguousarray(t.parent, np.int32), level=np.ascontiguousarray(t.level, np.int32),
                child=np.ascontiguousarray(t.child, np.int32), r=np.ascontiguousarray(t.r), x=np.ascontiguousarray(t.x))
    nf = _lib.NetFix()
    nf.n_bus, nf.slack, nf.n_levels, nf.max_children = t.n_bus, t.slack, t.n_levels, t.max_children
    nf.parent = keep["parent"].ctypes.data_as(C.POINTER(C.c_int32))
    nf.level = keep["level"].ctypes.data_as(C.POINTER(C.c_int32))
    nf.child = keep["child"].ctypes.data_as(C.POINTER(C.c_int32))
    nf.r = keep["r"].ctypes.data_as(C.POINTER(C.c_double))
    nf.x = keep["x"].ctypes.data_as(C.POINTER(C.c_double))
    pnet = pnet.contiguous()
    qnet = qnet.contiguous()
    n = pnet.shape[0]
    dev = pnet.device
    v = torch.empty_like(pnet)
    iters = torch.empty(n, dtype=torch.int32, device=dev)
    failed = torch.empty(n, dtype=torch.uint8, device=dev)
    isqr = pl = ql = None
    if want_branch:
        isqr, pl, ql = (torch.zeros_like(pnet) for _ in range(3))
    _lib.check(lib.pf_solve_batch(C.byref(nf), n, _ptr(pnet), _ptr(qnet), _ptr(v), _ptr(isqr), _ptr(pl), _ptr(ql),
                                  _ptr(iters), _ptr(failed), tol, max_iter, solver, _stream()),
               "pf_solve_batch")
    out = dict(v=v, iters=iters, failed=failed)
    if want_branch:
        out.update(isqr=isqr, pl=pl, ql=ql)
    return out


class FlexibilityProvisionEnv:
    """Drop-in for madrl/environments/flex_provision/flexibility_provision_env.py
    (``FlexibilityProvisionEnv(env_config_dict)``, train_agent.py:67).

    Same constructor argument, same reset()/step()/get_obs()/get_state()/get_avail_actions() and size
    getters, same dict-by-bus-id attributes safemaddpg.py reads, same _get_* accessors tester.py
    reads.  The episode draws come from the global NumPy RNG in the reference's order (env:85-87,
    100, 103) and are injected into the device env, so ``np.random.seed`` reproduces the reference's
    episode sequence on the same data.
    """

    def __init__(self, kwargs, net=None, series=None, device="cuda:0", **vec_kwargs):
        args = kwargs
        if isinstance(args, dict):
            merged = dict(DEFAULT_ENV_ARGS)
            merged.update(args)
            args = convert(merged)
        self.args = args                                                   # env:40
        self.model = getattr(self.args, "alg", None)                       # env:43
        self.data_path = args.data_path
        np.random.seed(args.seed)                                          # env:49
        self.vec = VecFlexProvisionEnv(self.args._asdict(), 1, device, net, series, **vec_kwargs)
        self.base_powergrid = self.vec.net                                 # env:52
        self.episode_limit = args.episode_limit
        self.action_space = ActionSpace(low=args.action_low, high=args.action_high)
        self.history = args.history
        self.n_agents = len(self.base_powergrid["buildings"])              # env:66
        self.n_actions = 4
        self.agent_ids = self.base_powergrid["buildings"]
        self.time_delta = self.vec.series.time_delta
        self._buses = list(self.base_powergrid["bus_numbers"])
        self._obs64 = torch.zeros(1, self.n_agents, self.vec.obs_size, dtype=torch.float64, device=self.vec.device)
        agents_obs, state = self.reset()                                   # env:69
        self.obs_size = agents_obs[0].shape[0]
        self.state_size = state.shape[0]

    # -- episode draws, reference order --------------------------------------------------
    def _select_start_hour(self):
        return np.random.choice(24)                                        # env:412

    def _select_start_day(self):
        return np.random.choice(self.vec.series.n_start_days(self.episode_limit))  # env:418-424

    def _select_start_interval(self):
        return np.random.choice(60 // self.time_delta)                     # env:416

    def get_action(self):                                                  # env:716-719
        return np.random.uniform(low=self.action_space.low, high=self.action_space.high,
                                 size=self.n_agents * self.n_actions)

    def _do_reset(self, day, hour, interval):
        e0 = np.array([np.random.uniform(0.9 * (self.args.e_max / 2), 1.1 * (self.args.e_max / 2))
                       for _ in self.base_powergrid["ESSs_at_buildings"]])  # env:100
        a0 = self.get_action()                                             # env:103
        spec = dict(day=[day], hour=[hour], interval=[interval], e0=e0[None], a0=a0[None])
        self.vec.reset(spec=spec, obs_out=self._obs64)
        return not bool(self.vec.failed.item())

    def reset(self):                                                       # env:74-155
        while True:
            self.start_hour = self._select_start_hour()
            self.start_day = self._select_start_day()
            self.start_interval = self._select_start_interval()
            if self._do_reset(self.start_day, self.start_hour, self.start_interval):
                break
            print("The power flow for the current initialization cannot be solved.")  # env:151
        self._refresh()
        return self._obs_list(), self.get_state()

    def manual_reset(self, day, hour, interval):                           # env:157-239
        self.start_day, self.start_hour, self.start_interval = day, hour, interval
        if not self._do_reset(day, hour, interval):
            # the reference retries the identical solve forever (env:214-237); fail loudly instead
            raise RuntimeError("The power flow for the current initialization cannot be solved.")
        self._refresh()
        return self._obs_list(), self.get_state()

    def step(self, actions):                                               # env:241-356
        a = np.asarray(actions).reshape(self.n_agents * self.n_actions)    # env:260
        t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(self.vec.device)
        reward, done, info = self.vec.step(t.view(1, self.n_agents, 4))
        vals = torch.cat([reward.view(1), info.view(-1), done.view(1).double(), self.vec.failed.view(1).double()]).cpu().numpy()
        rwd, terminated, failed = float(vals[0]), bool(vals[-2]), bool(vals[-1])
        info_d = {k: float(vals[1 + i]) for i, k in enumerate(_lib.INFO_KEYS)}
        if failed:
            print("The power flow for the current step cannot be solved.")  # env:315
            info_d["solver_failed"] = True                                 # env:337
        self._refresh()
        if terminated:
            print(f"Episode terminated at time: {self.steps} with return: {self.cumulative_reward:2.4f}.")  # env:351
        return rwd, terminated, info_d

    # -- mirrors of the reference's mutable attributes (safemaddpg.py:143-172,237,251) ---------
    # They are fetched from the device only when somebody reads one (tester.py, safemaddpg.py, run_env.py do;
    # the MADDPG training loop does not), so a plain step() costs one launch and one small device->host copy.
    _MIRRORED = ("current_active_demand", "current_reactive_demand", "current_pv_power", "current_price",
                 "current_voltage", "current_ess_energy", "initial_ess_energy", "power_reduction", "ess_charging",
                 "ess_discharging", "q_pv", "percentage_reduction", "steps", "cumulative_reward")

    def __getattr__(self, name):
        if name in FlexibilityProvisionEnv._MIRRORED and "vec" in self.__dict__:
            self._refresh_now()
            return self.__dict__[name]
        raise AttributeError(name)

    def _refresh(self):
        """Invalidate the mirrors; the next attribute read refetches them."""
        for k in FlexibilityProvisionEnv._MIRRORED:
            self.__dict__.pop(k, None)

    def _refresh_now(self):
        v = self.vec
        row = int(v.peek("ROW").item())
        data = v.series.table[row]
        nb, na = v.n_bus, v.n_agents
        self.current_active_demand = {b: data[i] for i, b in enumerate(self._buses)}          # env:617
        self.current_reactive_demand = {b: data[nb + i] for i, b in enumerate(self._buses)}   # env:618
        self.current_pv_power = {b: data[2 * nb + i] for i, b in enumerate(self.base_powergrid["PVs_at_buildings"])}
        self.current_price = np.array([data[2 * nb + na]])                                    # env:619
        blds = self.base_powergrid["buildings"]
        vm = v.peek("V").cpu().numpy()[0]
        self.current_voltage = {b: vm[i] for i, b in enumerate(self._buses)}
        for attr, key in (("current_ess_energy", "E"), ("initial_ess_energy", "E_INIT"), ("power_reduction", "PRED"),
                          ("ess_charging", "CH"), ("ess_discharging", "DIS"), ("q_pv", "QPV"),
                          ("percentage_reduction", "PCT")):
            arr = v.peek(key).cpu().numpy()[0]
            setattr(self, attr, {b: arr[i] for i, b in enumerate(blds)})
        self.steps = int(v.peek("STEPS").item())
        self.cumulative_reward = float(v.peek("CUMREW").item())

    def _obs_list(self):
        o = self._obs64.cpu().numpy()[0]
        return [o[i].copy() for i in range(self.n_agents)]

    def get_obs(self):                                                     # env:370-403 (stateful)
        self.vec.get_obs(obs_out=self._obs64)
        return self._obs_list()

    def get_obs_agent(self, agent_id):                                     # env:405-408
        return self.get_obs()[agent_id]

    def get_state(self):                                                   # env:358-368
        return self.vec.get_state().cpu().numpy()[0]

    # helpers safemaddpg.py calls on the env (safemaddpg.py:163-172)
    def _scale_and_clip_q_pv(self, reactive_action, active_power):         # env:621-626
        c = tan(acos(self.args.cos_phi_max)) * active_power
        return np.clip(-c + reactive_action * (c - (-c)), -c, c)

    def clip_percentage_reduction(self, percentage_reduction):             # env:676-677
        return {k: np.clip(v, 0, self.args.max_power_reduction) for k, v in percentage_reduction.items()}

    def adjust_ess_actions(self, ess_charging, ess_discharging):           # env:663-674
        for k in ess_charging:
            if ess_charging[k] > 0 and ess_discharging[k] > 0:
                if ess_charging[k] > ess_discharging[k]:
                    ess_charging[k] -= ess_discharging[k]
                    ess_discharging[k] = 0
                else:
                    ess_discharging[k] -= ess_charging[k]
                    ess_charging[k] = 0
        return ess_charging, ess_discharging

    def _clip_power_charging_discharging(self, charging, discharging, current_ess_energy):  # env:628-661
        a = self.args
        charging = np.clip(charging, 0, a.p_ch_max)
        discharging = np.clip(discharging, 0, a.p_dis_max)
        e_next = current_ess_energy + a.eta_ch * charging - (1 / a.eta_dis) * discharging
        if e_next > a.e_max:
            excess = e_next - a.e_max
            if charging > excess / a.eta_ch:
                charging -= excess / a.eta_ch
            else:
                discharging += (excess - charging * a.eta_ch) * a.eta_dis
                charging = 0
        elif e_next < a.e_min:
            lack = a.e_min - e_next
            if discharging > lack * a.eta_dis:
                discharging -= lack * a.eta_dis
            else:
                charging += (lack - discharging / a.eta_dis) / a.eta_ch
                discharging = 0
        return np.clip(charging, 0, a.p_ch_max), np.clip(discharging, 0, a.p_dis_max)

    # size getters / availability, env:708-738
    def get_obs_size(self):
        return self.obs_size

    def get_state_size(self):
        return self.state_size

    def get_avail_actions(self):
        return np.expand_dims(np.array([self.get_avail_agent_actions(i) for i in range(self.n_agents)]), axis=0)

    def get_avail_agent_actions(self, agent_id):
        return [1] * self.n_actions

    def get_total_actions(self):
        return self.n_actions

    def get_num_of_agents(self):
        return self.n_agents

    def close(self):
        self.vec.close()

    # tester.py accessors, env:740-778
    def _get_bus_v(self):
        return np.array([self.current_voltage[b] for b in self._buses])

    def _get_bus_active(self):
        return np.array([self.current_active_demand[b] for b in self._buses])

    def _get_bus_reactive(self):
        return np.array([self.current_reactive_demand[b] for b in self._buses])

    def _get_pv_active(self):
        return np.array([self.current_pv_power[b] for b in self.base_powergrid["PVs_at_buildings"]])

    def _get_pv_reactive(self):
        return np.array([self.q_pv[b] for b in self.base_powergrid["PVs_at_buildings"]])

    def _get_ess_energy(self):
        return np.array([self.current_ess_energy[b] for b in self.base_powergrid["ESSs_at_buildings"]])

    def _get_power_reduction(self):
        return np.array([self.power_reduction[b] for b in self.base_powergrid["buildings"]])

    def _get_ess_charging(self):
        return np.array([self.ess_charging[b] for b in self.base_powergrid["ESSs_at_buildings"]])

    def _get_ess_discharging(self):
        return np.array([self.ess_discharging[b] for b in self.base_powergrid["ESSs_at_buildings"]])

    def _get_price(self):
        return np.array([self.current_price])
