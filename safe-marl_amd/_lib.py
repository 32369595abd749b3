"""ctypes binding of include/flexenv.h.  The HIP library is the product path: if it is
missing or fails to load this module raises — there is no CPU fallback."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_NAME = "libflexenv_hip.so"
# FLEX_LIB_OVERRIDE: a diagnostic build of the same ABI (kernel variants for A/B timing, tools/*_variants.sh); never set in
# tests or in the product path — the digest check below is skipped for it
LIB_PATH = os.environ.get("FLEX_LIB_OVERRIDE") or os.path.join(_HERE, LIB_NAME)

FLEX_MAX_BUS = 64
FLEX_MAX_AGENTS = 8
FLEX_MAX_CHILDREN = 8
FLEX_INFO_W = 7
FLEX_F64, FLEX_F32 = 0, 1
FLEX_OK, FLEX_EINVAL, FLEX_ENOMEM, FLEX_EHIP = 0, -22, -12, -5
FLEX_ABI_VERSION = 2          # include/flexenv.h
FLEX_STEP_AUTORESET = 1
FLEX_STEP_OBS_RING = 16
FLEX_STEP_REPLAY_SINK = 4
FLEX_STEP_OBS_ROWS = 8
FLEX_STEP_MANY_NO_CARRY = 32
FLEX_ROW_FLOATS = 8
FLEX_SOLVER_TREE, FLEX_SOLVER_DENSE, FLEX_SOLVER_SWEEP = 0, 1, 2

PEEK = dict(V=0, E=1, E_INIT=2, PRED=3, CH=4, DIS=5, QPV=6, PCT=7, CUMREW=8, STEPS=9, ROW=10, START=11,
            PF_ITERS=12, EPISODE=13, PF_SWEEPS=14)
INFO_KEYS = ("reward", "revenue", "der_cost", "ess_cost", "discomfort_penalty", "voltage_penalty",
             "cumulative_reward")  # env:696-704


class FlexCfg(C.Structure):
    _fields_ = [
        ("n_agents", C.c_int32), ("history", C.c_int32), ("episode_limit", C.c_int32), ("per_hour", C.c_int32),
        ("n_start_days", C.c_int32), ("raw_actions", C.c_int32), ("pf_max_iter", C.c_int32), ("solver", C.c_int32),
        ("warm_start", C.c_int32), ("no_sweep_accel", C.c_int32),
        ("v_min", C.c_double), ("v_max", C.c_double), ("e_min", C.c_double), ("e_max", C.c_double),
        ("p_ch_max", C.c_double), ("p_dis_max", C.c_double), ("eta_ch", C.c_double), ("eta_dis", C.c_double),
        ("tan_phi", C.c_double), ("max_power_reduction", C.c_double), ("pv_cost", C.c_double),
        ("ess_cost", C.c_double), ("discomfort_coeff", C.c_double), ("voltage_coeff", C.c_double),
        ("dt", C.c_double), ("fail_penalty", C.c_double), ("pf_tol", C.c_double), ("action_low", C.c_double),
        ("action_high", C.c_double), ("seed", C.c_uint64),
    ]


class NetFix(C.Structure):
    _fields_ = [
        ("n_bus", C.c_int32), ("slack", C.c_int32), ("n_levels", C.c_int32), ("max_children", C.c_int32),
        ("parent", C.POINTER(C.c_int32)), ("level", C.POINTER(C.c_int32)), ("child", C.POINTER(C.c_int32)),
        ("r", C.POINTER(C.c_double)), ("x", C.POINTER(C.c_double)), ("agent_bus", C.POINTER(C.c_int32)),
    ]


class FlexReplaySink(C.Structure):
    """include/flexenv.h"""
    _fields_ = [(k, C.c_void_p) for k in ("policy_action", "hid_new", "small_ring", "hid_ring", "acc", "cursor_out",
                                          "aux_counter")] + \
               [(k, C.c_int32) for k in ("act_w", "hid_w", "small_w", "pad0")]


class SeriesTab(C.Structure):
    _fields_ = [("table", C.c_void_p), ("rows", C.c_int64), ("cols", C.c_int32)]


class ResetSpec(C.Structure):
    _fields_ = [("day", C.c_void_p), ("hour", C.c_void_p), ("interval", C.c_void_p), ("e0", C.c_void_p),
                ("a0", C.c_void_p)]


# every symbol include/flexenv.h declares
SYMBOLS = (
    "flexenv_create", "flexenv_destroy", "flexenv_reset", "flexenv_step", "flexenv_step_many", "flexenv_obs", "flexenv_obs_view", "flexenv_obs_source", "flexenv_state",
    "flexenv_peek", "flexenv_poke", "flexenv_num_envs", "flexenv_set_step_counter", "flexenv_set_obs_ring", "flexenv_set_replay_sink", "flexenv_rollout_burst", "flexenv_obs_size", "flexenv_state_size",
    "pf_solve_batch", "flexenv_safety_project", "flexenv_safety_project_env", "flexenv_version", "flexenv_abi_version",
    "flexnet_actor_forward", "flexnet_critic_tail_forward", "flexnet_critic_tail_backward", "flexnet_rollout_pack", "flexnet_wgrad", "flexnet_lnrelu_forward", "flexnet_lnrelu_backward", "flexnet_clip_rmsprop", "flexnet_clip_rmsprop_refresh", "flexnet_td_loss", "flexnet_td_stats", "flexnet_critic_td_backward", "flexnet_critic_td_backward_phases", "flexnet_wgrad_critic_finish",
    "flexnet_scaled_sum", "flexnet_agent_sum_explore", "flexnet_gather_rows", "flexnet_gather_rows_td", "flexnet_window_refresh", "flexnet_gather_window", "flexnet_linear2", "flexnet_gru_backward",
    "flexopf_qp_work_doubles", "flexopf_qp_solve",
)

class FlexActorArgs(C.Structure):
    """include/flexnet.h"""
    _fields_ = [("rows", C.c_int32), ("n_agents", C.c_int32), ("obs_dim", C.c_int32), ("act_dim", C.c_int32),
                ("agent_id", C.c_int32), ("layernorm", C.c_int32), ("ln_eps", C.c_float), ("variant", C.c_int32)] + \
               [(k, C.c_void_p) for k in ("obs", "hidden_in", "fc1_w", "fc1_b", "ln_w", "ln_b", "w_ih", "w_hh", "b_ih",
                                          "b_hh", "fc2_w", "fc2_b", "means", "hidden_out", "noise", "action", "env_action")] + \
               [("std", C.c_float), ("action_low", C.c_float), ("action_high", C.c_float), ("pad1", C.c_float),
                ("rng_state", C.c_void_p), ("cursor", C.c_void_p), ("obs_slab_stride", C.c_int64),
                ("hid_slab_stride", C.c_int64), ("cursor_out", C.c_void_p)] + \
               [(k, C.c_void_p) for k in ("save_z1", "save_x", "save_r", "save_z", "save_n", "save_hn")] + \
               [("ring_slabs", C.c_int64), ("obs_pushed", C.c_void_p), ("obs_row_stride", C.c_int32),
                ("obs_pushed_stride", C.c_int32), ("obs_slots", C.c_int32), ("obs_slot_w", C.c_int32)]


class FlexObsSource(C.Structure):
    """include/flexenv.h"""
    _fields_ = [("ring", C.c_void_p), ("pushed", C.c_void_p), ("row_stride", C.c_int32), ("pushed_stride", C.c_int32),
                ("slots", C.c_int32), ("slot_w", C.c_int32)]


class FlexWindowArgs(C.Structure):
    """include/flexnet.h"""
    _fields_ = [("row_ring", C.c_void_p), ("dst", C.c_void_p), ("rows", C.c_int64), ("first_slot", C.c_int64),
                ("n_envs", C.c_int32), ("n_agents", C.c_int32), ("history", C.c_int32), ("slabs", C.c_int32)]


class FlexQpArgs(C.Structure):
    """include/flexopf.h"""
    _fields_ = [(k, C.c_int32) for k in ("batch", "periods", "n_agents", "rows", "max_iter", "pad0")] + \
               [(k, C.c_double) for k in ("tol", "reg", "chain_a", "chain_b")] + \
               [(k, C.c_void_p) for k in ("q", "c", "lo", "hi", "free_mask", "jv", "v_lo", "v_hi", "ji", "i_hi", "e_lo", "e_hi",
                                          "x0", "x", "duals", "info", "work")]


FLEXOPF_INFO = 6


class FlexLinear2Args(C.Structure):
    """include/flexnet.h"""
    _fields_ = [("rows", C.c_int64)] + [(k, C.c_int32) for k in ("k1", "k2", "ld1", "ld2", "ldw", "c1", "c2", "pad0")] + \
               [(k, C.c_void_p) for k in ("x1", "x2", "w", "bias", "out", "x1_row_cell")]


class FlexBurstSafety(C.Structure):
    """include/flexenv.h"""
    _fields_ = [("s_p", C.c_void_p), ("s_q", C.c_void_p), ("beta", C.c_void_p), ("v_min", C.c_double), ("v_max", C.c_double),
                ("penalty", C.c_double), ("adjusted", C.c_void_p), ("env_action", C.c_void_p), ("act_low", C.c_float),
                ("act_high", C.c_float)]


class FlexGruBwdArgs(C.Structure):
    """include/flexnet.h"""
    _fields_ = [("rows", C.c_int32), ("act_dim", C.c_int32)] + \
               [(k, C.c_void_p) for k in ("d_means", "d_hidden", "fc2_w", "r", "z", "n", "hn", "h_prev", "d_gi", "d_gh",
                                          "w_ih", "z1", "x", "fc1_w", "fc1_b", "ln_w", "ln_b", "dz", "d_ln_w", "d_ln_b", "d_fc1_b",
                                          "d_id", "workspace")] + \
               [("workspace_floats", C.c_int64), ("d_id_agent_stride", C.c_int64), ("d_id_unit_stride", C.c_int64),
                ("fc1_ld", C.c_int32), ("obs_dim", C.c_int32), ("n_agents", C.c_int32), ("agent_id", C.c_int32),
                ("layernorm", C.c_int32), ("ln_eps", C.c_float)]


class FlexCriticTailArgs(C.Structure):
    """include/flexnet.h"""
    _fields_ = [("rows", C.c_int32), ("layernorm", C.c_int32), ("ln_eps", C.c_float), ("variant", C.c_int32)] + \
               [(k, C.c_void_p) for k in ("z1", "ln_w", "ln_b", "fc2_w", "fc2_b", "fc3_w", "fc3_b", "q", "dq", "dz1",
                                          "d_ln_w", "d_ln_b", "d_fc2_w", "d_fc2_b", "d_fc3_w", "d_fc3_b", "z_shared", "z_id")] + \
               [("n_agents", C.c_int32), ("overwrite_grads", C.c_int32), ("d_z_shared", C.c_void_p), ("d_z_id", C.c_void_p),
                ("workspace", C.c_void_p), ("workspace_floats", C.c_int64),
                ("d_z_id_agent_stride", C.c_int32), ("d_z_id_unit_stride", C.c_int32),
                ("z_id_agent_stride", C.c_int32), ("z_id_unit_stride", C.c_int32),
                ("dq_uniform", C.c_int32), ("dq_value", C.c_float), ("q_mean_scale", C.c_float), ("variant_pgrad32", C.c_int32),
                ("q_mean_out", C.c_void_p)]


FLEXNET_CRITIC_WS_FLOATS = 1024 * 4416


class FlexWgradArgs(C.Structure):
    _fields_ = [("k", C.c_int64), ("lda", C.c_int64), ("ldb", C.c_int64), ("workspace_floats", C.c_int64),
                ("m", C.c_int32), ("n", C.c_int32), ("accumulate", C.c_int32), ("ldc", C.c_int32),
                ("a", C.c_void_p), ("b", C.c_void_p), ("c", C.c_void_p), ("workspace", C.c_void_p),
                ("colsum", C.c_void_p), ("b2", C.c_void_p), ("c2", C.c_void_p), ("ldb2", C.c_int64), ("n2", C.c_int32),
                ("ldc2", C.c_int32), ("b_row_cell", C.c_void_p)]


FLEXNET_WGRAD_WS_FLOATS = 520 * 12288 + 520 * 192


class FlexLnReluArgs(C.Structure):
    _fields_ = [("rows", C.c_int32), ("n_agents", C.c_int32), ("layernorm", C.c_int32), ("ln_eps", C.c_float)] + \
               [(n, C.c_void_p) for n in ("z", "bias", "id_cols", "ln_w", "ln_b", "out", "dout", "dz", "d_bias", "d_id",
                                          "d_ln_w", "d_ln_b", "workspace")] + [("workspace_floats", C.c_int64)]


FLEXNET_LNRELU_WS_FLOATS = 1024 * 704


FLEXNET_OPT_MAX_TENSORS = 16
FLEXNET_OPT_MAX_ELEMENTS = 1 << 20


class FlexClipRmspropArgs(C.Structure):
    _fields_ = [("n_tensors", C.c_int32), ("lr", C.c_float), ("alpha", C.c_float), ("eps", C.c_float),
                ("max_norm", C.c_float), ("pad0", C.c_int32), ("total_norm", C.c_void_p), ("workspace", C.c_void_p),
                ("numel", C.c_int64 * FLEXNET_OPT_MAX_TENSORS), ("param", C.c_void_p * FLEXNET_OPT_MAX_TENSORS),
                ("grad", C.c_void_p * FLEXNET_OPT_MAX_TENSORS), ("square_avg", C.c_void_p * FLEXNET_OPT_MAX_TENSORS),
                ("step", C.c_void_p * FLEXNET_OPT_MAX_TENSORS)]


class FlexTdLossArgs(C.Structure):
    """include/flexnet.h"""
    _fields_ = [("rows", C.c_int32), ("n_agents", C.c_int32), ("normalise", C.c_int32), ("gamma", C.c_float),
                ("bn_eps", C.c_float), ("bn_momentum", C.c_float)] + \
               [(k, C.c_void_p) for k in ("reward", "done", "next_q", "q", "bn_weight", "bn_bias", "running_mean",
                                          "running_var", "num_batches_tracked", "dq", "loss", "workspace")] + \
               [("workspace_floats", C.c_int64), ("stats_ready", C.c_int32), ("pad0", C.c_int32), ("stat_rows", C.c_int64)]


FLEXNET_TD_WS_FLOATS = 2 * (64 * 2 * 8 + 1024)
FLEXNET_TD_STAT_DOUBLES = 64 * 2 * 8


class FlexAgentSumArgs(C.Structure):
    """include/flexnet.h"""
    _fields_ = [("n_envs", C.c_int32), ("n_agents", C.c_int32), ("act_dim", C.c_int32), ("pad0", C.c_int32),
                ("act_low", C.c_float), ("act_high", C.c_float)] + \
               [(k, C.c_void_p) for k in ("means", "eps", "std", "action", "env_action")]


class FlexSumArgs(C.Structure):
    """include/flexnet.h"""
    _fields_ = [("n", C.c_int64), ("scale", C.c_float), ("pad0", C.c_int32), ("x", C.c_void_p), ("out", C.c_void_p),
                ("workspace", C.c_void_p), ("workspace_floats", C.c_int64)]


FLEXNET_SUM_WS_FLOATS = 2 * 64


class FlexRolloutPackArgs(C.Structure):
    """include/flexnet.h"""
    _fields_ = [(k, C.c_int32) for k in ("n_envs", "n_agents", "obs_dim", "act_dim", "slabs", "small_w", "info_w",
                                         "cursor_stepped")] + \
               [(k, C.c_void_p) for k in ("action", "reward", "obs_next", "done", "hid_new", "info", "failed", "obs_ring",
                                          "hid_ring", "small_ring", "hid_state", "cursor", "info_sum", "rew_sum", "fail_sum",
                                          "rng_state")]


FLEXNET_GATHER_MAX_JOBS = 12


class FlexGatherArgs(C.Structure):
    """include/flexnet.h"""
    _fields_ = [("n_jobs", C.c_int32), ("pad0", C.c_int32),
                ("src", C.c_void_p * FLEXNET_GATHER_MAX_JOBS), ("dst", C.c_void_p * FLEXNET_GATHER_MAX_JOBS),
                ("rows", C.c_int64 * FLEXNET_GATHER_MAX_JOBS), ("width", C.c_int32 * FLEXNET_GATHER_MAX_JOBS),
                ("src_stride", C.c_int32 * FLEXNET_GATHER_MAX_JOBS), ("dst_stride", C.c_int32 * FLEXNET_GATHER_MAX_JOBS)]


FLEXNET_WINDOW_MAX_JOBS = 8
FLEXNET_WINDOW_MAX_CELLS = 4


class FlexWindowRefreshArgs(C.Structure):
    """include/flexnet.h"""
    _fields_ = [("n_jobs", C.c_int32), ("n_cells", C.c_int32), ("start", C.c_void_p), ("ring_rows", C.c_int64),
                ("base", C.c_void_p * FLEXNET_WINDOW_MAX_JOBS), ("dst", C.c_void_p * FLEXNET_WINDOW_MAX_JOBS),
                ("rows", C.c_int64 * FLEXNET_WINDOW_MAX_JOBS), ("row_off", C.c_int64 * FLEXNET_WINDOW_MAX_JOBS),
                ("width", C.c_int32 * FLEXNET_WINDOW_MAX_JOBS), ("src_stride", C.c_int32 * FLEXNET_WINDOW_MAX_JOBS),
                ("cell", C.c_void_p * FLEXNET_WINDOW_MAX_CELLS), ("cell_mod", C.c_int64 * FLEXNET_WINDOW_MAX_CELLS),
                ("reward_job", C.c_int32), ("pad0", C.c_int32)]


FLEXNET_EUNSUPPORTED = -3

_lib = None


class FlexLibraryError(RuntimeError):
    pass


def load():
    """Load libflexenv_hip.so (built in-tree by safe_marl_amd.build / __graft_entry__.build)."""
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own libamdhip64 (same SONAME as /opt/rocm's).  It must be mapped FIRST so that
    # this library resolves to the same HIP runtime instance torch allocates device memory with;
    # loading in the other order leaves two runtimes in the process and every pointer foreign.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise FlexLibraryError(
            f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the product path.")
    # the binary must be THIS tree's: its build stamp is the content hash of sources, headers and flags (build.py).  A
    # binary from other sources is rebuilt in place when the compiler is here (it is on the GPU box), refused otherwise.
    if os.environ.get("FLEX_SKIP_DIGEST_CHECK") != "1" and not os.environ.get("FLEX_LIB_OVERRIDE"):
        from . import build as _build
        built, want = _build.built_digest(), _build.source_digest()
        if built != want:
            if os.path.exists(_build.HIPCC):
                import warnings
                warnings.warn(f"{LIB_NAME} was built from other sources (stamp {str(built)[:16]}, tree {want[:16]}): rebuilding")
                _build.build(force=True)
            else:
                raise FlexLibraryError(
                    f"{LIB_PATH} was built from other sources (stamp {str(built)[:16]}, tree {want[:16]}) and there is no "
                    f"{_build.HIPCC} to rebuild it: run `python -c 'import __graft_entry__ as g; g.build()'`")
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as exc:  # pragma: no cover - depends on the box
        raise FlexLibraryError(f"cannot load {LIB_PATH}: {exc}") from exc
    missing = [s for s in SYMBOLS if not hasattr(lib, s)]
    if missing:
        raise FlexLibraryError(f"{LIB_PATH} lacks symbols {missing}")
    vp, i32 = C.c_void_p, C.c_int32
    lib.flexenv_create.argtypes = [C.POINTER(FlexCfg), C.POINTER(NetFix), C.POINTER(SeriesTab), i32, i32,
                                   C.POINTER(vp)]
    lib.flexenv_create.restype = C.c_int
    lib.flexnet_actor_forward.argtypes = [C.POINTER(FlexActorArgs), vp]
    lib.flexnet_actor_forward.restype = C.c_int
    lib.flexnet_rollout_pack.argtypes = [C.POINTER(FlexRolloutPackArgs), vp]
    lib.flexnet_rollout_pack.restype = C.c_int
    lib.flexnet_gather_rows.argtypes = [C.POINTER(FlexGatherArgs), vp]
    lib.flexnet_gather_rows.restype = C.c_int
    lib.flexnet_gather_rows_td.argtypes = [C.POINTER(FlexGatherArgs), i32, i32, C.POINTER(FlexTdLossArgs), vp]
    lib.flexnet_gather_rows_td.restype = C.c_int
    lib.flexnet_window_refresh.argtypes = [C.POINTER(FlexWindowRefreshArgs), C.POINTER(FlexTdLossArgs), vp]
    lib.flexnet_window_refresh.restype = C.c_int
    lib.flexnet_gru_backward.argtypes = [C.POINTER(FlexGruBwdArgs), vp]
    lib.flexnet_gru_backward.restype = C.c_int
    lib.flexenv_set_step_counter.argtypes = [vp, vp, C.c_int64]
    lib.flexenv_set_step_counter.restype = C.c_int
    lib.flexenv_set_obs_ring.argtypes = [vp, vp, C.c_int64, i32]
    lib.flexenv_set_obs_ring.restype = C.c_int
    lib.flexenv_set_replay_sink.argtypes = [vp, C.POINTER(FlexReplaySink)]
    lib.flexenv_set_replay_sink.restype = C.c_int
    lib.flexenv_rollout_burst.argtypes = [vp, C.POINTER(FlexActorArgs), vp, vp, vp, vp, vp, i32, C.POINTER(FlexBurstSafety), vp]
    lib.flexenv_rollout_burst.restype = C.c_int
    lib.flexnet_critic_td_backward.argtypes = [C.POINTER(FlexCriticTailArgs), C.POINTER(FlexTdLossArgs), vp]
    lib.flexnet_critic_td_backward.restype = C.c_int
    lib.flexnet_critic_td_backward_phases.argtypes = [C.POINTER(FlexCriticTailArgs), C.POINTER(FlexTdLossArgs), i32, vp]
    lib.flexnet_critic_td_backward_phases.restype = C.c_int
    lib.flexnet_wgrad_critic_finish.argtypes = [C.POINTER(FlexWgradArgs), C.POINTER(FlexCriticTailArgs), C.POINTER(FlexTdLossArgs), vp]
    lib.flexnet_wgrad_critic_finish.restype = C.c_int
    lib.flexnet_wgrad.argtypes = [C.POINTER(FlexWgradArgs), vp]
    lib.flexnet_wgrad.restype = C.c_int
    lib.flexnet_clip_rmsprop.argtypes = [C.POINTER(FlexClipRmspropArgs), vp]
    lib.flexnet_clip_rmsprop.restype = C.c_int
    lib.flexnet_clip_rmsprop_refresh.argtypes = [C.POINTER(FlexClipRmspropArgs), C.POINTER(FlexWindowRefreshArgs), C.POINTER(FlexTdLossArgs), vp]
    lib.flexnet_clip_rmsprop_refresh.restype = C.c_int
    lib.flexnet_td_loss.argtypes = [C.POINTER(FlexTdLossArgs), vp]
    lib.flexnet_td_loss.restype = C.c_int
    lib.flexnet_td_stats.argtypes = [C.POINTER(FlexTdLossArgs), vp]
    lib.flexnet_td_stats.restype = C.c_int
    lib.flexnet_agent_sum_explore.argtypes = [C.POINTER(FlexAgentSumArgs), vp]
    lib.flexnet_agent_sum_explore.restype = C.c_int
    lib.flexnet_scaled_sum.argtypes = [C.POINTER(FlexSumArgs), vp]
    lib.flexnet_scaled_sum.restype = C.c_int
    for fn in (lib.flexnet_lnrelu_forward, lib.flexnet_lnrelu_backward):
        fn.argtypes = [C.POINTER(FlexLnReluArgs), vp]
        fn.restype = C.c_int
    for fn in (lib.flexnet_critic_tail_forward, lib.flexnet_critic_tail_backward):
        fn.argtypes = [C.POINTER(FlexCriticTailArgs), vp]
        fn.restype = C.c_int
    lib.flexenv_destroy.argtypes = [vp]
    lib.flexenv_destroy.restype = None
    lib.flexenv_reset.argtypes = [vp, vp, C.POINTER(ResetSpec), vp, i32, vp, vp]
    lib.flexenv_reset.restype = C.c_int
    lib.flexenv_step.argtypes = [vp, vp, i32, vp, vp, vp, vp, vp, i32, i32, vp]
    lib.flexenv_step.restype = C.c_int
    lib.flexenv_step_many.argtypes = [vp, vp, i32, i32, i32, vp, vp, vp, vp, i32, vp]
    lib.flexenv_step_many.restype = C.c_int
    lib.flexenv_obs.argtypes = [vp, vp, i32, vp]
    lib.flexenv_obs.restype = C.c_int
    lib.flexenv_obs_view.argtypes = [vp, vp, i32, vp]
    lib.flexenv_obs_view.restype = C.c_int
    lib.flexenv_obs_source.argtypes = [vp, C.POINTER(FlexObsSource)]
    lib.flexenv_obs_source.restype = C.c_int
    lib.flexnet_gather_window.argtypes = [C.POINTER(FlexWindowArgs), vp]
    lib.flexnet_gather_window.restype = C.c_int
    lib.flexnet_linear2.argtypes = [C.POINTER(FlexLinear2Args), vp]
    lib.flexnet_linear2.restype = C.c_int
    lib.flexopf_qp_work_doubles.argtypes = [i32, i32, i32]
    lib.flexopf_qp_work_doubles.restype = C.c_int64
    lib.flexopf_qp_solve.argtypes = [C.POINTER(FlexQpArgs), vp]
    lib.flexopf_qp_solve.restype = C.c_int
    lib.flexenv_state.argtypes = [vp, vp, vp]
    lib.flexenv_state.restype = C.c_int
    lib.flexenv_peek.argtypes = [vp, i32, vp, vp]
    lib.flexenv_peek.restype = C.c_int
    lib.flexenv_poke.argtypes = [vp, i32, vp, vp]
    lib.flexenv_poke.restype = C.c_int
    for name in ("flexenv_num_envs", "flexenv_obs_size", "flexenv_state_size"):
        getattr(lib, name).argtypes = [vp]
        getattr(lib, name).restype = i32
    lib.pf_solve_batch.argtypes = [C.POINTER(NetFix), i32, vp, vp, vp, vp, vp, vp, vp, vp, C.c_double, i32, i32, vp]
    lib.pf_solve_batch.restype = C.c_int
    lib.flexenv_safety_project.argtypes = [vp, vp, i32, vp, vp, vp, C.c_double, C.c_double, C.c_double, vp, vp, vp]
    lib.flexenv_safety_project.restype = C.c_int
    lib.flexenv_safety_project_env.argtypes = [vp, vp, i32, vp, vp, vp, C.c_double, C.c_double, C.c_double, vp, vp,
                                               C.c_float, C.c_float, vp, vp]
    lib.flexenv_safety_project_env.restype = C.c_int
    lib.flexenv_version.argtypes = []
    lib.flexenv_version.restype = C.c_char_p
    lib.flexenv_abi_version.argtypes = []
    lib.flexenv_abi_version.restype = i32
    if lib.flexenv_abi_version() != FLEX_ABI_VERSION:
        raise FlexLibraryError(f"libflexenv_hip.so speaks ABI {lib.flexenv_abi_version()}, this binding ABI {FLEX_ABI_VERSION} "
                               "(include/flexenv.h): rebuild with safe_marl_amd.build.build(force=True)")
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        raise FlexLibraryError(f"{what} failed with code {rc}")
