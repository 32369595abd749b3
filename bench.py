#!/usr/bin/env python3
"""bench.py — env-steps/s of the batched flexibility-provision step on MI355X.

A "step" is ONE vector step of the hot path over one batch of synthetic input: for every one of the
4096 environments of this GPU, the step body (action parse -> 33-bus AC power flow -> ESS update ->
reward, with the `get_obs()` that always follows it fused in, and the restart of the environments that just
terminated).  The K steps are issued as launches of `flexenv_step_many` (up to --steps-per-launch steps each, the vectorised
run_env.py:78-92; the driver's K = 20 is ONE launch); `single_launch_sibling` in the same line is the one-launch-per-step form
(`flexenv_step`).  Inputs (series table, action pool) are resident in HBM before the timed region starts; nothing crosses
PCIe inside it (tools/pcie_probe.py has the host-buffer rates).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Environments are independent, so ranks share nothing on the data path of the headline metric (weak scaling, no
collective); the only collectives there are the contract's barrier and the max-over-ranks of the elapsed time.  Under
N > 1 the line also carries `value_device_events` (max over ranks of the HIP-event time of each rank's K steps: the figure without
the closing barrier's own collective), what the process group reports about itself (`dist`), and per training leg the all-reduce
counts against the schedule and `replica_max_abs_diff`; training legs that have not finished after --train-deadline seconds are
given up and the line is printed without them.

Next to the headline the same JSON line carries:
  * `sustained`: the same env-only step over >= 2048 steps in this process, run FIRST (the device enters the timed region at its
    operating clocks);
  * `train`: the TRAINING loop of model.py:198-267 + model.py:40-71 — rollout (policy inference + env step + replay
    write) and the 11 gradient steps per 60 vector steps — for BASELINE configs 3 (MADDPG, 5 and 3 agents, 4096 envs) and
    4 (SAFEMADDPG, 8192 envs); with N > 1 ranks config 5: MADDPG at 4096 envs per GPU with the flat gradient bucket
    all-reduced through RCCL before every clip + RMSprop step (`dist.allreduce_flat` between the two HIP graphs of a
    sub-update);
  * `train_kernel_shares`: per-kernel GPU time of one MADDPG training episode (torch.profiler in a child process started
    before this process touches the GPU; null when it is unavailable, e.g. under rocprofv3).

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself
(`python -m torch.distributed.run ...` as a child process, before anything touches the GPU) and relays rank 0's line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_ENVS = 4096           # BASELINE.json: 4096 envs per GPU
ACTION_POOL = 16        # distinct action tensors cycled through, all resident in HBM
# algorithmic bytes per env-step of the fused PF/step kernel (fp64):
#   reads 784 B (actions 160 + series row 576 + E 40 + step/start 8), writes 556 B (V 264 + E 40 +
#   reward 8 + done/failed 4 + new obs features 240) = 1340 B; + 2880 B when the stacked fp32
#   observation [5,144] is materialised for the learner, which the fused kernel does = 4220 B.
B_ALG_CORE = 1340
B_ALG_WITH_OBS = 4220
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2048)
    ap.add_argument("--warmup", type=int, default=256)
    ap.add_argument("--envs", type=int, default=N_ENVS, help="envs per GPU (the metric is quoted at 4096)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train", action="store_true", help="skip the training legs (`train`)")
    ap.add_argument("--no-sustained", action="store_true", help="skip the >= 2048-step env-only leg (`sustained`)")
    ap.add_argument("--no-kernel-shares", action="store_true", help="skip the torch.profiler child (`train_kernel_shares`)")
    ap.add_argument("--train-deadline", type=int, default=150,
                    help="N > 1: seconds after which unfinished training legs are given up and the line is printed without them "
                         "(below the process group's 180 s collective timeout; 0 = off)")
    ap.add_argument("--train-episodes", type=int, default=12,
                    help="timed 95-step episodes per training leg (>= 2).  12 episodes = 1140 vector steps = lcm(95, 60): exactly "
                         "19 update events, the cadence's long-run average (3 episodes hold 4 or 5 events depending on phase)")
    ap.add_argument("--strict", action="store_true",
                    help="exit non-zero when a training leg fails or runs without its rollout graph / both update graphs")
    ap.add_argument("--kernel-shares-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-seconds", type=float, default=1.0, help="stepping time of ONE CPU baseline sample (C3: 5 samples per solver after 1 s of warm-up, C2: 3; x threads = CPU work)")
    ap.add_argument("--warm-start", type=int, default=1)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL; gloo only to "
                    "rehearse the multi-rank control flow on a one-GPU box together with FLEX_BENCH_ONE_DEVICE=1)")
    ap.add_argument("--pf-tol", type=float, default=1e-12, help="power-flow convergence threshold (inf-norm power "
                    "mismatch, pu); 1e-12 is the headline setting, the parity bar is 1e-6 on voltages and rewards")
    ap.add_argument("--no-sweep-accel", action="store_true", help="plain sweeps: without the two-sweep extrapolation (A/B leg)")
    ap.add_argument("--no-graph", action="store_true", help="launch every step eagerly instead of replaying a HIP graph")
    ap.add_argument("--stacked-obs", action="store_true",
                    help="get_obs() as a stacked [n_agents, 6 * history] copy per step (the round-3 kernel) instead of the row push")
    ap.add_argument("--launch-form", choices=["many", "single"], default="many",
                    help="many: the K steps as launches of --steps-per-launch steps each (flexenv_step_many: every wavefront walks "
                         "its own environments through the sequence, run_env.py:78-92 vectorised); single: one flexenv_step launch "
                         "per step, replayed as HIP graphs (the headline form of rounds 1-4; `single_launch_sibling` either way)")
    ap.add_argument("--steps-per-launch", type=int, default=256, help="launch length of --launch-form many (a shorter region is ONE launch)")
    ap.add_argument("--solver", choices=["sweep", "newton"], default="sweep",
                    help="sweep: backward/forward sweeps + Newton verification; newton: NR with tree elimination")
    return ap.parse_args()


def usable_cores():
    """Host cores this process may actually use: the affinity mask capped by the cgroup CPU quota (the GPU box
    shows 256 CPUs but grants 16; 256 OpenMP threads on a 16-CPU quota run 10x slower than 16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cpu_baseline(net, series, seconds, envs=4096, samples=5, warmup_s=1.0):
    """The oracle's CPU restatements timed on this box's host cores (rank 0, N=1 only), following BASELINE.md §3:

      C1  NumPy / Python, ONE env, one core — the structural analogue of the reference (oracle/env_oracle.py);
      C2  the C restatement (oracle/flexenv_oracle.c), one thread, `envs` environments;
      C3  the same with OpenMP over the environments on every core the cgroup grants, pinned.

    C2 / C3 are timed with BOTH of the oracle's solvers: `dense_nr` (polar Newton-Raphson, dense LU: O(n^3) per iteration — the
    checker the parity tests use) and `distflow_sweep` (backward/forward sweep in the reference's own DistFlow variables: O(n)
    per iteration, the class of algorithm the HIP kernel runs).  `value` is C3 with the FASTER of the two — the like-for-like
    figure; the dense one sits beside it.  Every figure is the median of `samples` samples of >=
    `seconds` s of stepping (resets outside the timed region) after `warmup_s` s of warm-up; `spread` = (max - min) / median."""
    import numpy as np
    cores = usable_cores()
    os.environ["OMP_NUM_THREADS"] = str(cores)
    os.environ.setdefault("OMP_PROC_BIND", "close")
    os.environ.setdefault("OMP_PLACES", "cores")
    os.environ.setdefault("OMP_DYNAMIC", "false")
    from oracle import c_oracle
    from oracle.env_oracle import FlexEnvOracle
    rng = np.random.default_rng(1234)
    n = int(envs)
    acts = rng.uniform(0.5, 1.0, (8, n, 5, 4))
    busy_total = [0.0]

    def time_c(solver, threads, min_s, n_samples, warm):
        got_threads = c_oracle.set_threads(threads)        # (libgomp may have read its environment long ago: set it by call)
        env = c_oracle.COracleEnv(net, series.table, n, solver=solver)

        def fresh():
            day = rng.integers(0, series.n_start_days(96), n)
            start = rng.integers(0, 4, n) + rng.integers(0, 24, n) * 4 + day * 96
            env.reset(start, rng.uniform(0.01125, 0.01375, (n, 5)), rng.uniform(0, 1, (n, 20)))

        state = {"k": 90}

        def sample(min_t):
            busy, steps = 0.0, 0
            while busy < min_t:
                if state["k"] >= 90:                       # a whole episode is used up: restart outside the timed steps
                    fresh()
                    state["k"] = 0
                t0 = time.perf_counter()
                env.step(acts[state["k"] % 8])
                busy += time.perf_counter() - t0
                state["k"] += 1
                steps += 1
            busy_total[0] += busy * got_threads
            return n * steps / busy

        if warm > 0:
            sample(warm)
        rates = sorted(sample(min_s) for _ in range(n_samples))
        med = rates[len(rates) // 2]
        return {"value": med, "threads": got_threads, "envs": n, "samples": [round(r) for r in rates],
                "spread": (rates[-1] - rates[0]) / med}

    out_c = {}
    for solver in ("distflow_sweep", "dense_nr"):
        out_c[solver] = {"C3": time_c(solver, cores, seconds, samples, warmup_s),
                         "C2": time_c(solver, 1, seconds, 3, 0.3)}
    # C1: one Python environment, scalar loop
    one = FlexEnvOracle(net, {}, series.active, series.reactive, series.pv, series.price)
    c1_rates = []
    a1 = rng.uniform(0.5, 1.0, (8, 5, 4))
    for _ in range(3):
        one.reset()
        t0 = time.perf_counter()
        k = 0
        while k < 90 and time.perf_counter() - t0 < max(0.5, seconds / 2):
            one.step(a1[k % 8]); one.get_obs(); k += 1
        dt1 = time.perf_counter() - t0
        busy_total[0] += dt1
        c1_rates.append(k / dt1)
    c1_rates.sort()
    best = max(out_c, key=lambda sv: out_c[sv]["C3"]["value"])
    algo = {"distflow_sweep": "DistFlow backward/forward sweep, O(n)/iteration (reference variables pf.py:65-94), cold start",
            "dense_nr": "polar Newton-Raphson on the dense Ybus, O(n^3) LU/iteration"}
    main = out_c[best]["C3"]
    return {"value": main["value"], "unit": "env-steps/s", "cores": main["threads"], "kind": "port", "algo": algo[best],
            "envs": n, "samples": main["samples"], "spread": main["spread"],
            "sample": (f"C3 of BASELINE.md §3: median of {len(main['samples'])} x >= {seconds:g} s of step()+get_obs() (stacked copy), "
                       f"{n} envs, oracle/flexenv_oracle.c, OpenMP, {main['threads']} pinned threads; {busy_total[0]:.0f} core-seconds "
                       "in all legs; kind 'port': Pyomo + IPOPT (utils/pf.py:101-102) cannot run on this box (BASELINE.md §2)"),
            "obs_form": "stacked [5, 144] copy per step — the GPU leg with the same observation form is `stacked_sibling`",
            "C1_numpy_one_env": {"value": c1_rates[len(c1_rates) // 2], "threads": 1, "envs": 1, "algo": algo["dense_nr"] + " (NumPy)",
                                 "what": "oracle/env_oracle.py, one Python env, scalar loop: the structural analogue of the reference"},
            "C2_one_thread": {k: out_c[k]["C2"] for k in out_c},
            "C3_all_cores": {k: out_c[k]["C3"] for k in out_c},
            "algos": algo}


TRAIN_ALG_ARGS = dict(  # madrl/args/default.yaml merged with alg_args/maddpg.yaml (examples/train_maddpg.py)
    gumbel_softmax=False, epsilon_softmax=False, softmax_eps=None, episodic=False, cuda=True, grad_clip_eps=1.0,
    save_model_freq=40, replay_warmup=0, policy_lrate=1.0e-4, value_lrate=1.0e-4, mixer_lrate=None, target=True,
    target_lr=0.1, entr=1.0e-3, max_steps=240, batch_size=32, replay=True, replay_buffer_size=5.0e3, agent_type="rnn",
    agent_id=True, shared_params=True, layernorm=True, mixer=False, gaussian_policy=False, LOG_STD_MIN=0.0,
    LOG_STD_MAX=0.5, fixed_policy_std=1.0, hid_activation="relu", init_type="normal", init_std=0.1,
    action_enforcebound=True, double_q=True, clip_c=1.0, gamma=0.99, hid_size=64, continuous=True,
    normalize_advantages=False, train_episodes_num=400, behaviour_update_freq=60, target_update_freq=120,
    policy_update_epochs=1, value_update_epochs=10, mixer_update_epochs=None, reward_normalisation=True, eval_freq=20,
    num_eval_episodes=10, action_low=0, action_high=1.0, action_bias=0.0, action_scale=1.0,
)


def make_trainer(alg, n_agents, envs, rank, local_rank, batch_div=4, intended=False):
    """PGTrainer(args, model, env, logger) as train_agent.py:67-107 builds it, on the vectorised HIP env.  A sub-update's batch
    is batch_size x (envs / batch_div) consecutive replay slots: batch_div = 4 is the trainer's default (1.47 samples consumed
    per transition collected), batch_div = 1 the reference's own sample reuse (5.87 = 11 x 32 / 60, model.py:43-50 x
    replay_buffer.py:17-21)."""
    import torch
    from safe_marl_amd.flex_env import VecFlexProvisionEnv
    from safe_marl_amd.learner import MADDPG, SAFEMADDPG
    from safe_marl_amd.network import create_network
    from safe_marl_amd.series import make_synthetic_series
    from safe_marl_amd.trainer import PGTrainer
    from safe_marl_amd.util import convert
    blds = [5, 10, 15, 20, 25] if n_agents == 5 else [5, 15, 25]
    env_args = {"buildings": blds, "pv_nodes": blds, "ess_nodes": blds}
    if alg == "safemaddpg":
        env_args["alg"] = "safemaddpg"
    net = create_network(env_args)
    series = make_synthetic_series(net, n_days=365)
    env = VecFlexProvisionEnv(env_args, envs, device=f"cuda:{local_rank}", net=net, series=series,
                              seed=1234 + 1000 * rank, warm_start=True)
    d = dict(TRAIN_ALG_ARGS)
    d.update(alg=alg, agent_num=env.n_agents, obs_size=env.obs_size, state_size=env.state_size, action_dim=4,
             v_min=0.9, v_max=1.1)
    torch.manual_seed(0)
    trainer = PGTrainer(convert(d), {"maddpg": MADDPG, "safemaddpg": SAFEMADDPG}[alg], env, None,
                        batch_scale=max(1, envs // batch_div), replay_capacity=envs * 96 * 2)
    if intended:
        # NOT the reference's routing (under it SAFEMADDPG's policy cannot move the environment): the safety layer's
        # output reaches env.step() as the physical values of building i, agent-major (learner.SAFEMADDPG.intended_actions)
        trainer.behaviour_net.intended_actions = True
    return trainer, env


def train_leg(alg, n_agents, envs, episodes, rank, local_rank, world, barrier, max_over_ranks, batch_div=4, intended=False):
    """One training configuration: TWO warm-up episodes (allocations, HIP-graph captures of the rollout, its bursts and both
    sub-updates — the first update event falls into the first episode, the first policy/value replays and the 8/4/2-step
    bursts of the second window into the second —, rocBLAS plans), then `episodes` timed episodes of 95 vector steps each
    between barriers; whole-job env-steps/s = envs x world x steps / max-rank time.  `episode_ms` lists the timed episodes
    one by one (an episode holds one or two update events, so they alternate)."""
    import torch
    trainer, env = make_trainer(alg, n_agents, envs, rank, local_rank, batch_div, intended=intended)
    stat = {}
    for _ in range(2):
        trainer.behaviour_net.train_process(stat, trainer)
    barrier()
    from safe_marl_amd import dist as fdist
    ar0 = dict(fdist.STATS)
    steps0 = trainer.steps
    episode_ms = []
    t0 = time.perf_counter()
    for _ in range(episodes):
        t1 = time.perf_counter()
        trainer.behaviour_net.train_process(stat, trainer)      # (ends on a device -> host read of the episode statistics)
        episode_ms.append((time.perf_counter() - t1) * 1e3)
    torch.cuda.synchronize()
    dt_local = time.perf_counter() - t0
    barrier()
    dt = max_over_ranks(dt_local)
    steps = trainer.steps - steps0
    freq = trainer.args.behaviour_update_freq
    events = sum(1 for st in range(steps0, trainer.steps) if st > 0 and st % freq == 0)
    per_event = trainer.args.value_update_epochs + trainer.args.policy_update_epochs
    # the exchange step, checkable from the line itself: what THIS rank put through the all-reduce in the
    # timed region against what the schedule asks for (one flat bucket per gradient step), and whether the replicas still
    # hold the same weights afterwards — max over ranks of |theta - theta_rank0|, which must be exactly 0.0
    ar_calls = fdist.STATS["allreduce_calls"] - ar0["allreduce_calls"]
    ar_bytes = fdist.STATS["allreduce_bytes"] - ar0["allreduce_bytes"]
    diverge = fdist.replica_divergence(trainer.behaviour_net) if world > 1 else {"params": 0.0, "buffers": 0.0}
    rg = getattr(trainer.behaviour_net, "_rollout_graph", None)
    # the rollout alone, after the timed region (the leg is over: the extra steps go nowhere): runs of 60 vector steps, the
    # distance between two update events, as the training loop issues them (one burst launch, or graphs of 16 / 8 / 4 bodies)
    rollout_us = None
    if rg is not None and rg.graph is not None:
        rg.run(60)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(5):
            rg.run(60)
        e1.record()
        torch.cuda.synchronize()
        rollout_us = e0.elapsed_time(e1) * 1e3 / 300
    out = {"alg": alg, "action_routing": ("intended (NOT the reference's routing; learner.SAFEMADDPG.intended_actions)" if intended else
                                          "reference (bug-compatible, SURVEY A13)") if alg == "safemaddpg" else None,
           "n_agents": env.n_agents, "envs_per_gpu": envs, "n_gpus": world, "episodes": episodes,
           "vector_steps": steps, "ms_per_vector_step": dt / steps * 1e3,
           "env_steps_per_s": envs * world * steps / dt, "grad_steps": events * per_event,
           "batch_per_gpu": trainer.effective_batch_size(),
           # samples a gradient step consumes per transition the rollout collects: (10 + 1) x batch per 60 x envs
           "samples_per_transition": per_event * trainer.effective_batch_size() / float(freq * envs),
           "episode_ms": [round(x, 3) for x in episode_ms],
           "rollout_graph": bool(rg is not None and rg.graph is not None), "rollout_fused": bool(rg is not None and rg.fast),
           # policy + environment for a whole run of steps in ONE persistent launch (flexenv_rollout_burst; plain MADDPG)
           "rollout_burst_launch": bool(rg is not None and rg.fused_burst),
           "rollout_us_per_vector_step": None if rollout_us is None else round(rollout_us, 2),
           # update events whose value sub-updates read Q'(s', pi(s')) filed once for the union of their windows
           # (trainer.replay_event; chosen per event where the windows overlap enough — the reference's sample reuse)
           "bootstrap_cached_events": int(getattr(trainer, "bootstrap_cached_events", 0)),
           "graphed_updates": sorted(trainer._update_graphs), "split_update_graphs": bool(world > 1 and trainer._update_graphs and
                                        not all(g.get("allreduce_in_graph") for g in trainer._update_graphs.values())),
           "allreduce_in_graph": bool(world > 1 and trainer._update_graphs and
                                      all(g.get("allreduce_in_graph") for g in trainer._update_graphs.values())),
           "grad_allreduce": (f"{torch.distributed.get_backend()} all-reduce of one flat bucket (sum, 1/world inside graph B), "
                              "before the clip") if world > 1 else None,
           "allreduce_calls": ar_calls if world > 1 else None, "allreduce_calls_expected": events * per_event if world > 1 else None,
           "allreduce_bytes": ar_bytes if world > 1 else None, "graph_form_agreements": fdist.STATS["agreements"] if world > 1 else None,
           "replica_max_abs_diff": diverge["params"] if world > 1 else None,
           "replica_buffers_max_abs_diff": diverge["buffers"] if world > 1 else None,
           "reward_bn_statistics": ("cross-rank" if getattr(trainer, "sync_reward_bn", False) else "per-rank (buffers differ by design)") if world > 1 else None,
           "mean_train_reward": float(stat.get("mean_train_reward", float("nan"))),
           "mean_train_value_loss": float(stat.get("mean_train_value_loss", float("nan")))}
    del trainer, env, rg
    import gc
    gc.collect()                  # the leg's HIP graphs die here, not inside the next leg's capture (util.graph_capture)
    torch.cuda.empty_cache()
    return out


def learner_rooflines():
    """The learner-side kernels against THEIR rooflines, timed live with HIP events on torch's current stream (the stream
    these kernels are launched on): the fused actor forward on the matrix cores (exact fp32 MFMA, 157.3 TFLOP/s dense
    peak, MI355X_MICROARCH.md) at the update batch and at the rollout batch, the batch-reduced weight gradient and the
    replay-window gather against HBM (8 TB/s).  Algorithmic flops / bytes are stated per entry."""
    import torch
    from safe_marl_amd.nets import RNNAgent, fused_actor_forward, tall_wgrad
    from safe_marl_amd.replay_buffer import TransReplayBuffer
    from safe_marl_amd.util import convert
    d = dict(TRAIN_ALG_ARGS)
    d.update(agent_num=5, obs_size=144, action_dim=4)
    agent = RNNAgent(149, convert(d)).cuda()

    def timed(fn, n=100):
        for _ in range(10):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e-3                       # seconds per launch

    def graph_timed(fn, reps=20, n=20):
        """Per-call time inside a HIP graph of `reps` calls (how the rollout issues it): an eager call costs the host ~20 us
        of Python + ctypes, more than a rollout-size kernel takes."""
        from safe_marl_amd.util import graph_capture
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                fn()
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with graph_capture(g):
            for _ in range(reps):
                fn()
        for _ in range(3):
            g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(n):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / (n * reps) * 1e-3

    out = []
    for b, what in ((32768, "update batch: bootstrap actions of a value sub-update (eager launches)"),
                    (4096, "rollout batch: one vector step (HIP-graph replays of 20 calls)")):
        obs = torch.randn(b, 5, 144, device="cuda")
        hid = torch.randn(b, 5, 64, device="cuda")
        with torch.no_grad():
            t = (timed if b > 4096 else graph_timed)(lambda: fused_actor_forward(agent, obs, hid, 5, True))
        flops = 2.0 * b * 5 * (149 * 64 + 2 * 192 * 64 + 64 * 4)
        out.append({"kernel": "actor_forward_mfma_kernel (32-row tiles)" if b * 5 > 20480 else
                    "actor_rollout16_kernel (five 16-row tiles per CU)", "rows": b * 5, "what": what, "bound": "mfma", "dtype": "f32",
                    "achieved": flops / t / 1e12, "peak": 157.3, "unit": "TFLOP/s", "frac": flops / t / 1e12 / 157.3,
                    "launch_us": t * 1e6})
    k, m, n = 32768, 64, 720
    dy, x = torch.randn(k, m, device="cuda"), torch.randn(k, n, device="cuda")
    t = timed(lambda: tall_wgrad(dy, x))
    out.append({"kernel": "wgrad_kernel<2,5> + reduce", "shape": [k, m, n], "what": "critic fc1 weight gradient dY^T X", "bound": "hbm",
                "achieved": 4.0 * k * (m + n) / t / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": 4.0 * k * (m + n) / t / 1e9 / HBM_PEAK_GBS, "launch_us": t * 1e6,
                "note": "algorithmic bytes = both operands once; the kernel sits at the HBM / fp32-MFMA balance point"})
    out.append(critic_backward_roofline(timed))
    N, bs = 4096, 32768
    buf = TransReplayBuffer(N * 24, device="cuda")
    buf.alloc_slabs(N, 5, 144, 4, 64, history=24)          # row mode: the layout the fused rollout writes
    buf.row_ring.normal_()
    buf.row_ring.view(buf.slabs, N, 5, buf.ROW_W)[..., 6] = 23.0
    buf.k, buf.first = 40, 23
    win = torch.zeros(bs + N, 720, device="cuda")
    hidw = torch.zeros(bs, 320, device="cuda")
    plan = [(buf.obs_source_ring, 0, None, 0, bs + N, win), ("hid_ring", 0, None, N, bs, hidw)]
    t = timed(lambda: buf.gather(plan, 25 * N + 17))
    # algorithmic bytes: every feature row of the window's reach once in (window + 23 + 1 slabs of 32-byte records), the
    # stacked static batch out, the hidden block in and out
    moved = 4.0 * ((bs + N) * 720 + 2 * bs * 320) + 32.0 * 5 * (bs + N + 23 * N)
    out.append({"kernel": "gather_window_kernel + gather_rows_kernel",
                "what": "replay window -> static batch of a value sub-update: stacked observations formed from the row ring "
                        "(the gather is the im2col), hidden states copied", "bound": "hbm",
                "achieved": moved / t / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": moved / t / 1e9 / HBM_PEAK_GBS,
                "launch_us": t * 1e6, "note": "bytes read (each row record once) + written"})
    return out


def critic_backward_roofline(timed):
    """flexnet_critic_td_backward at the update batch (32 768 samples x 5 agents): reward statistics, the matrix-core backward
    that forms q / the TD error / dLoss/dq itself, the dz1 fold and the one-launch finish.  fp32 MFMA work per row: forward
    recompute 2 * 64 * 64, dz2 -> da1 2 * 64 * 64, dW2 2 * 64 * 64 flops (+ the 64-wide fc3 / LayerNorm vector work, not
    counted)."""
    import ctypes as C
    import torch
    import torch.nn as nn
    from safe_marl_amd import _lib
    from safe_marl_amd.nets import MLPCritic, _critic_args, _critic_workspace, _td_args
    from safe_marl_amd.util import convert
    d = dict(TRAIN_ALG_ARGS)
    d.update(agent_num=5, obs_size=144, action_dim=4)
    c = MLPCritic(745, 1, convert(d)).cuda()
    lib = _lib.load()
    samples, n = 32768, 5
    shared, ids = torch.randn(samples, 64, device="cuda"), torch.randn(n, 64, device="cuda")
    nq, rew = torch.randn(samples, n, device="cuda"), torch.randn(samples, n, device="cuda")
    done = (torch.rand(samples, device="cuda") < 0.05).float()
    bn = nn.BatchNorm1d(n).cuda().train()
    ws = _critic_workspace(shared.device)
    grads = torch.empty(64 * 64 + 64 * 4 + 1, device="cuda")
    dz1 = torch.empty(samples * n, 64, device="cuda")
    d_shared, d_id, loss = torch.empty_like(shared), torch.empty(n, 64, device="cuda"), torch.empty((), device="cuda")
    a = _critic_args(shared, c.layernorm.weight, c.layernorm.bias, c.fc2.weight, c.fc2.bias, c.fc3.weight, c.fc3.bias,
                     c.layernorm.eps)
    a.rows, a.z1, a.z_shared, a.z_id, a.n_agents = samples * n, None, shared.data_ptr(), ids.data_ptr(), n
    a.dz1 = dz1.data_ptr()
    a.d_fc2_w, a.d_fc2_b, a.d_fc3_w = grads.data_ptr(), grads[4096:].data_ptr(), grads[4160:].data_ptr()
    a.d_ln_w, a.d_ln_b, a.d_fc3_b = grads[4224:].data_ptr(), grads[4288:].data_ptr(), grads[4352:].data_ptr()
    a.workspace, a.workspace_floats, a.overwrite_grads = ws.data_ptr(), ws.numel(), 1
    a.d_z_shared, a.d_z_id = d_shared.data_ptr(), d_id.data_ptr()
    t_args = _td_args(rew, done, nq, 0.99, bn)
    t_args.loss = loss.data_ptr()
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def call():
        _lib.check(lib.flexnet_critic_td_backward(C.byref(a), C.byref(t_args), stream), "flexnet_critic_td_backward")

    t = timed(call)
    flops = 3.0 * 2 * 64 * 64 * samples * n
    return {"kernel": "critic_tail_pgrad16_kernel<TD, SM> + statistics, finish", "rows": samples * n,
            "what": "value loss + critic backward as flexnet_critic_td_backward launches them (3 launches; inside a captured value "
                    "sub-update the statistics ride in the batch refresh and the finish in the weight gradient's second stage)",
            "bound": "mfma", "dtype": "f32",
            "achieved": flops / t / 1e12, "peak": 157.3, "unit": "TFLOP/s", "frac": flops / t / 1e12 / 157.3, "launch_us": t * 1e6,
            "note": "16-row tiles on v_mfma_f32_16x16x4_f32, two wavefronts per SIMD; sample-major: d_z_shared / d_z_id formed in the "
                    "backward kernel, no dz1 round trip, no fold launch (3 launches: statistics, backward, finish)"}


def kernel_shares_child():
    """Child process (started before the parent touches the GPU): per-kernel GPU time of one MADDPG training episode
    (5 agents, 4096 envs) through torch.profiler; prints one JSON line."""
    import torch
    import safe_marl_amd  # noqa: F401
    from torch.profiler import ProfilerActivity, profile
    trainer, env = make_trainer("maddpg", 5, N_ENVS, 0, 0)
    stat = {}
    trainer.behaviour_net.train_process(stat, trainer)          # warm-up: graph captures
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        trainer.behaviour_net.train_process(stat, trainer)
        torch.cuda.synchronize()
    agg = {}
    for ev in prof.key_averages():
        if "cuda" not in str(getattr(ev, "device_type", "")).lower():
            continue                                            # CPU-side rows repeat their kernels' time
        t = float(getattr(ev, "self_device_time_total", 0.0) or 0.0)
        if t <= 0:
            continue
        name = ev.key
        cut = name.find("(")
        name = name if cut < 0 else name[:cut]
        e = agg.setdefault(name, [0.0, 0])
        e[0] += t
        e[1] += int(ev.count)
    total = sum(v[0] for v in agg.values())
    if total <= 0:
        raise SystemExit("no device activity recorded")
    top = sorted(agg.items(), key=lambda kv: -kv[1][0])[:16]
    subs = sum(v[1] for k, v in agg.items() if "clip_rmsprop_kernel" in k)      # one optimiser step per sub-update
    print(json.dumps({"episode": f"MADDPG 5 agents x 4096 envs, one episode of 95 vector steps incl. {subs} sub-updates "
                                 f"(an update event every 60 steps: 10 value + 1 policy)",
                      "gpu_ms_total": total / 1e3,
                      "top": [{"kernel": k[:96], "calls": v[1], "ms": v[0] / 1e3, "share": v[0] / total} for k, v in top]}),
          flush=True)


def spawn_kernel_shares():
    """Run kernel_shares_child() in a child started BEFORE this process initialises the GPU; None on any failure."""
    import subprocess
    if any("rocprof" in (k + "=" + v).lower() for k, v in os.environ.items()):
        # a profiler is attached to this process tree (rocprofv3 exports ROCPROF_* and preloads its tool library): its
        # preloaded library may already have initialised the GPU in THIS process, and a child started from such a process
        # is exactly the exec the GPU boxes forbid — no child, no kernel shares
        return None
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--kernel-shares-child"], capture_output=True,
                           text=True, timeout=240)
        for line in reversed(r.stdout.strip().splitlines()):
            if line.startswith("{"):
                return json.loads(line)
    except Exception as exc:
        print(f"[bench] kernel-share child failed: {exc}", file=sys.stderr)
    return None


def launch_ranks(a):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a CHILD process tree (never exec from a
    process that has touched the GPU — this one has not) and relay rank 0's JSON line."""
    import socket
    import subprocess
    if any("rocprof" in (k + "=" + v).lower() for k, v in os.environ.items()):
        raise SystemExit("bench.py --gpus N under a profiler: start the ranks with torch.distributed.run yourself (a process the "
                         "profiler has attached to must not start children that use the GPU)")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    r = subprocess.run(cmd)
    sys.exit(r.returncode)


# SURVEY.md 8(d), tree-structured solve: algorithmic floating-point operations per env-step, counted from the sweep's
# arithmetic (csrc/flex_device.h pf_sweep / zbus_apply): per bus and sweep ~60 (current conj(S/V) 10, subtree scan 5 x 2 + 4,
# line drop 6, path scan 5 x 2 fma + 2, voltage update 2, mismatch test 8, re/im bookkeeping), 32 buses, x sweeps per
# solve (measured: pf_sweeps_mean), + the fp64 Ybus verification (~50 per bus) + actions / reward / ESS (~400 per env)
def flops_per_env_step(sweeps_mean):
    return 32 * 60 * sweeps_mean + 32 * 50 + 400


def roofline_details(a, env, kern_ms):
    """What bounds the dominant kernel, from the build and from counters (FLOP/s, waves per SIMD, register /
    LDS occupancy next to the bandwidth figure): the compiler's resource report of the library being timed
    (safe-marl_amd/kernel_resources.json, written by build.py), and the committed counter passes of tools/env_counters.sh
    (profiles/pmc_traffic.json) — used only when they were taken on THIS build (source digest) at this batch size;
    otherwise `traffic`, `valu_issue_frac` and the counter-derived `limiter` are null and `counters_stale` says why."""
    from safe_marl_amd import build
    out = {"traffic": None, "kernel": None, "limiter": None}
    # the instantiation flexenv_step launches for this configuration (csrc/flexenv.hip: EPW, ObsT, ActT, NA_CAP, SINK)
    epw = 2 if env.n_bus - 1 <= 32 else 1
    lw = 64 // epw
    small = env.n_agents == 5 and 3 * env.history <= (3 if epw == 2 else 2) * lw
    many = a.launch_form == "many" and not a.stacked_obs
    want = (f"flex_step_many_kernel<{epw}, float>" if many else
            f"flex_step_kernel<{epw}, float, float, {5 if small else 8}, false, false>" if a.stacked_obs else
            f"flex_step_kernel<{epw}, float, float, 5, false, true>")
    res = [v for v in build.kernel_resources("flex_step_many_kernel<" if many else "flex_step_kernel<").values() if v.get("name") == want]
    out["kernel"] = want
    waves = (a.envs + epw - 1) // epw
    out["waves_launched"] = waves
    out["resident_waves_per_simd"] = waves / 1024.0                  # 256 CUs x 4 SIMDs
    if res:
        r = res[0]
        out.update(vgprs=r.get("vgprs"), agprs=r.get("agprs"), sgprs=r.get("sgprs"),
                   scratch_bytes_per_lane=r.get("scratch_bytes_per_lane"), lds_bytes_per_block=r.get("lds_bytes_per_block"),
                   max_waves_per_simd=r.get("waves_per_simd"))
    sweeps = float(env.peek("PF_SWEEPS").float().mean().item())
    fl = flops_per_env_step(sweeps)
    out["flops_per_env_step"] = round(fl)
    out["achieved_gflops"] = fl * a.envs / (kern_ms * 1e-3) / 1e9      # mixed fp64 / fp32 (fp64 vector peak: 78.6 TFLOP/s)
    # (the headline form's passes; the single-launch kernel's sit beside them: tools/env_counters.sh with FORM=single)
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json" if many else "pmc_traffic_single_launch.json")
    try:
        t = json.load(open(tpath))
    except Exception:
        out["counters_stale"] = os.path.basename(tpath) + " missing under profiles/"
        return out
    if t.get("source_digest") != build.built_digest():
        out["counters_stale"] = "counter passes were taken on another build (source digest differs)"
        return out
    if int(t.get("envs_per_launch", -1)) != a.envs or a.solver != "sweep" or want not in (t.get("kernel") or []):
        out["counters_stale"] = "counter passes were taken at another batch size / solver"
        return out
    # counters are per LAUNCH of the counter passes (`steps_per_launch` steps each); the line quotes them per launch of THIS run
    spl_t = int(t.get("steps_per_launch", 1))
    spl = a.steps_per_launch_used if many else 1
    out["counter_steps_per_launch"] = spl_t
    scale = spl / float(spl_t)
    out["traffic"] = int(t.get("flex_step_kernel_bytes_per_launch") * scale) if t.get("flex_step_kernel_bytes_per_launch") else None
    out["traffic_per_env_step"] = (t.get("flex_step_kernel_bytes_per_launch") / spl_t / a.envs) if t.get("flex_step_kernel_bytes_per_launch") else None
    c = t.get("counters_per_launch", {})
    wave_q, valu_q = c.get("SQ_WAVE_CYCLES"), c.get("SQ_ACTIVE_INST_VALU")
    if wave_q and valu_q:
        gui = c.get("GRBM_GUI_ACTIVE")
        kern_cycles = gui / 8.0 if gui else None                    # summed over the 8 XCDs (MI355X_MICROARCH.md)
        out["valu_insts_per_wave"] = c.get("SQ_INSTS_VALU", 0) / max(1.0, c.get("SQ_WAVES", waves))
        # SQ_ACTIVE_INST_* / SQ_WAIT_* / SQ_WAVE_CYCLES count quad-cycles (same guide): fractions of a wavefront's lifetime
        f_valu = valu_q / wave_q
        f_mem = c.get("SQ_WAIT_ANY", 0.0) / wave_q
        f_dep = c.get("SQ_WAIT_INST_ANY", 0.0) / wave_q
        out["wave_cycle_split"] = {"valu_issue": round(f_valu, 3), "other_issue": round(max(0.0, c.get("SQ_ACTIVE_INST_ANY", valu_q) - valu_q) / wave_q, 3),
                                   "waiting_on_memory_or_lds": round(f_mem, 3), "issue_stalled": round(f_dep, 3)}
        if kern_cycles:
            # VALU-issue cycles of ALL the SIMD's wavefronts over the SIMD-cycles of the launch (1024 SIMDs)
            out["valu_issue_frac"] = 4.0 * valu_q / (1024.0 * kern_cycles)
            out["kernel_cycles"] = round(kern_cycles)
        per_simd = out["resident_waves_per_simd"]
        parts = sorted((("VALU issue", f_valu), ("waits on memory / LDS", f_mem), ("issue stalls (dependencies)", f_dep)),
                       key=lambda kv: -kv[1])
        out["limiter"] = ("%s (%.0f %% of wavefront cycles; then %s %.0f %%, %s %.0f %%) at %.1f resident wavefronts per SIMD; not "
                          "HBM bandwidth: counter traffic is %.1f MB per launch" % (
                              parts[0][0], 100 * parts[0][1], parts[1][0], 100 * parts[1][1], parts[2][0], 100 * parts[2][1],
                              per_simd, (out["traffic"] or 0) / 1e6))
    return out


def main():
    a = parse_args()
    if a.kernel_shares_child:
        return kernel_shares_child()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(a)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # (FLEX_BENCH_FORCE_DIST=1: the process-group code paths — RCCL init with a device id, the gathered device report, barriers, the
    #  maximum over ranks, shutdown — with ONE rank under torch.distributed.run, all a one-GPU box allows of the real backend)
    distributed = world > 1 or os.environ.get("FLEX_BENCH_FORCE_DIST") == "1"
    if a.gpus != world:
        print(f"[bench] --gpus {a.gpus} but WORLD_SIZE={world}: reporting n_gpus = {world}", file=sys.stderr)
    shares = None
    if rank == 0 and not distributed and not a.no_train and not a.no_kernel_shares:
        shares = spawn_kernel_shares()                          # before this process touches the GPU

    import torch
    import safe_marl_amd  # noqa: F401
    from safe_marl_amd.network import create_network
    from safe_marl_amd.series import make_synthetic_series
    from safe_marl_amd.flex_env import VecFlexProvisionEnv

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    if os.environ.get("FLEX_BENCH_ONE_DEVICE") == "1":       # rehearsal only: every rank on cuda:0
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import datetime
        # a collective that cannot complete (a rank died) errors out after three minutes instead of hanging the job
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(seconds=180))
        else:
            dist.init_process_group(a.backend, timeout=datetime.timedelta(seconds=180))
    coll_dev = dev if a.backend == "nccl" else torch.device("cpu")
    dist_info = None
    if distributed:
        # what the process group itself reports, and which device every rank sits on (N ranks on N distinct devices, or the
        # one-device rehearsal) — gathered, not assumed
        mine = {"rank": rank, "device_index": torch.cuda.current_device(), "device": torch.cuda.get_device_name(),
                "uuid": str(getattr(torch.cuda.get_device_properties(torch.cuda.current_device()), "uuid", ""))}
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        dist_info = {"world_size": dist.get_world_size(), "backend": str(dist.get_backend()),
                     "distinct_devices": len({(g["device_index"], g["uuid"]) for g in gathered}), "ranks": gathered}

    net = create_network()
    series = make_synthetic_series(net)                      # 1096 days x 96 rows x 72 cols fp64 (60.6 MB)
    env = VecFlexProvisionEnv({}, a.envs, device=f"cuda:{local_rank}", net=net, series=series,
                              seed=1234 + 1000 * rank, warm_start=bool(a.warm_start), pf_tol=a.pf_tol,
                              solver={"sweep": 2, "newton": 0}[a.solver], sweep_accel=not a.no_sweep_accel)
    gen = torch.Generator(device=dev)
    gen.manual_seed(99 + rank)
    # the range the reference's translate_action actually delivers (util.py:125-128), float32 like util.py:184
    pool = (0.5 + 0.5 * torch.rand(ACTION_POOL, a.envs, env.n_agents, 4, device=dev, generator=gen)).float()
    env.reset()

    def one_step(k):
        # ONE launch: step + get_obs; envs that terminate restart inside the same launch (FLEX_STEP_AUTORESET).  get_obs() is
        # the ROW PUSH of round 4 (FLEX_STEP_OBS_ROWS): the step appends its 6-feature row per agent to the env's history, where
        # the policy kernels read the stacked [5, 144] observation in place; `--stacked-obs` times the round-3 form (a stacked
        # copy per step) instead, and `stacked_sibling` in the JSON line carries that figure either way
        if a.stacked_obs:
            env.step(pool[k % ACTION_POOL], fuse_obs=True, auto_reset=True)
        else:
            env.step(pool[k % ACTION_POOL], obs_rows=True, auto_reset=True)

    # --launch-form many (round 5, the default): the K steps as launches of L = --steps-per-launch steps (flexenv_step_many —
    # the vectorised form of the reference's open-loop runner, run_env.py:78-92: a given action sequence, step(), per-step
    # records), a region shorter than L as ONE launch, a remainder as one more.  Every step does what one_step's launch does —
    # same loads, stores and arithmetic, bit-identical results (tests/test_step_many_gpu.py) — and writes its own row of
    # reward / done / info / failed; what is gone is the launch boundary per step.  Step k of a launch reads pool[k mod 16].
    many = a.launch_form == "many" and not a.stacked_obs
    a.steps_per_launch_used = 1
    many_out = {}

    def many_launch(n_steps):
        # (arguments checked and marshalled once per launch length — VecFlexProvisionEnv.step_many_prepared, whose first call
        #  also runs the launch once, untimed; every later call is the bare flexenv_step_many call on the same buffers)
        if n_steps not in many_out:
            out_ = (torch.empty(n_steps, a.envs, dtype=torch.float64, device=dev),
                    torch.empty(n_steps, a.envs, dtype=torch.uint8, device=dev),
                    torch.empty(n_steps, a.envs, 7, dtype=torch.float64, device=dev),
                    torch.empty(n_steps, a.envs, dtype=torch.uint8, device=dev))
            many_out[n_steps] = env.step_many_prepared(pool, steps=n_steps, auto_reset=True, out=out_)[0]
            return
        many_out[n_steps]()

    if many:
        L_ = max(1, a.steps_per_launch)
        a.steps_per_launch_used = min(L_, a.steps)
        for n_ in {L_, a.steps % L_, a.warmup % L_, min(L_, a.steps)} - {0}:     # buffers allocated and every form launched once, untimed
            many_launch(n_)
        torch.cuda.synchronize()

    # The loop is launch-issue sensitive (9 us of Python + ctypes per launch against a 14 us kernel), so consecutive steps
    # are captured as HIP graphs and replayed: blocks of ACTION_POOL steps plus ONE graph for the remainder (K mod 16); a
    # short run (the driver's --steps 20) is one graph of exactly K step launches.  Work per step is identical either way (one flexenv_step
    # launch per step); `--no-graph` keeps everything eager.
    graphs = {}

    from safe_marl_amd.util import graph_capture

    def capture(n_steps):
        g = torch.cuda.CUDAGraph()
        with graph_capture(g):
            for j in range(n_steps):
                one_step(j)
        return g

    if not a.no_graph and not many:
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                one_step(0)
            torch.cuda.current_stream().wait_stream(side)
            sizes = {ACTION_POOL, a.steps % ACTION_POOL, a.warmup % ACTION_POOL}
            if a.steps <= 128:
                sizes.add(a.steps)                    # a short timed region (the driver's --steps 20) is ONE graph launch
            for n_steps in sorted(sizes - {0}):
                graphs[n_steps] = capture(n_steps)
            # A graph's FIRST launch uploads it to the device (measured, tools/launch_region_probe.py: ~40 us once, i.e. 2 us
            # per step of a 20-step region, against 242-252 us for every later launch of the same graph).  Instantiation is
            # setup like the capture itself: every graph is launched once here, untimed, as hipGraphUpload would do (PyTorch
            # does not expose it).  The W warm-up steps and the K timed steps follow unchanged.
            for g in graphs.values():
                g.replay()
            torch.cuda.synchronize()
        except Exception as exc:                      # capture not available: eager launches
            print(f"[bench] HIP graph capture failed ({exc}); eager launches", file=sys.stderr)
            graphs = {}

    def run_steps(count):
        done_steps = 0
        if many:
            L = max(1, a.steps_per_launch)
            for _ in range(count // L):
                many_launch(L)
            if count % L:
                many_launch(count % L)
            return
        if count in graphs and count != ACTION_POOL:
            graphs[count].replay()
            return
        if ACTION_POOL in graphs:
            while count - done_steps >= ACTION_POOL:
                graphs[ACTION_POOL].replay()
                done_steps += ACTION_POOL
        rest = count - done_steps
        if rest and rest in graphs:
            graphs[rest].replay()
            done_steps += rest
        for k in range(done_steps, count):
            one_step(k)

    bar_cell = torch.zeros(1, device=dev) if distributed and a.backend == "nccl" else None

    def barrier():
        if distributed:
            if bar_cell is not None:
                # on RCCL a barrier IS an all-reduce of one element followed by a synchronize (ProcessGroupNCCL::barrier); issued on a
                # cell allocated once it does without the fill launch dist.barrier() queues for a fresh tensor every time — behind
                # the 20-step launch: +5.0 us instead of +10.7 (tools/barrier_probe.py, profiles/r05dd_barrier_probe.txt)
                dist.all_reduce(bar_cell)
            else:
                dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if not distributed:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # `sustained`: the same step over >= 2048 launches, HIP events on the launch stream around the whole region.  It runs FIRST
    # (round 5): the device then enters the W warm-up steps and the K timed steps at its operating clocks, as it would in any
    # use where environment steps follow one another.  Measured (tools/launch_seq_probe.py, profiles/r05w_launch_seq_probe.txt):
    # the 20-step region the driver asks for takes 245-252 us on a busy device and 283-293 us on one that idled for 50-500 ms
    # of host-side set-up (W = 5 steps = 60 us do not bring the clocks back); `config.timed_after` says so in the line.
    sustained = None
    if not a.no_sustained:
        n_sus = max(2048, ACTION_POOL * (a.steps // ACTION_POOL))
        barrier()
        s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t1 = time.perf_counter()
        s0.record()
        run_steps(n_sus)
        s1.record()
        barrier()
        sus_elapsed = max_over_ranks(time.perf_counter() - t1)
        sustained = {"steps": n_sus, "value": a.envs * world * n_sus / sus_elapsed, "unit": "env-steps/s",
                     "ms_per_step": sus_elapsed / n_sus * 1e3, "device_ms_per_step": s0.elapsed_time(s1) / n_sus}

    run_steps(a.warmup)                                   # W untimed warm-up steps

    barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()                                          # (on an idle stream: done before the first step is enqueued)
    t0 = time.perf_counter()
    run_steps(a.steps)                                    # exactly K steps
    ev1.record()
    # closing bracket, as the contract words it: barrier + synchronize, THEN the clock; the job's time is the MAX over ranks.  Under
    # N > 1 that puts the closing barrier's own collective (one small all-reduce queued behind the steps: tens of microseconds
    # across eight devices) inside a region the driver's K = 20 makes 0.15 ms long — apparatus, not workload: the ranks share
    # nothing on this path.  The line therefore also carries the maximum over the ranks of each rank's DEVICE time for its K steps
    # (the HIP events ev0 / ev1 on the launch stream: `value_device_events`), beside the headline, never instead of it.
    barrier()
    elapsed = max_over_ranks(time.perf_counter() - t0)
    dev_ms = ev0.elapsed_time(ev1)
    dev_ms_max = max_over_ranks(dev_ms)

    # roofline leg.  The timed region above is K back-to-back launches of ONE kernel (flex_step_kernel) on one
    # stream, bracketed by the HIP events ev0/ev1 on that stream: dev_ms / K is its average launch duration
    # (rocprofv3 --kernel-trace --stats of the same command agrees: profiles/).  A second pass brackets every
    # launch with its own event pair; that figure carries ~2 us of event overhead per launch and is reported
    # as `bracketed_launch_ms` only.
    n_ev = 12 if many else min(max(a.steps, 64), 400)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_ev)]
    torch.cuda.synchronize()
    for k, (s, e) in enumerate(evs):
        s.record()
        if many:
            many_launch(a.steps_per_launch_used)          # (a launch of the timed region's length)
        else:
            one_step(k)
        e.record()
    torch.cuda.synchronize()
    durs = sorted(s.elapsed_time(e) for s, e in evs)
    # the kernel's average launch duration: from the longer of the two event-bracketed regions (a 20-step region is
    # 0.3 ms, where the event pair's own ~5 us shows)
    kern_ms = sustained["device_ms_per_step"] if sustained is not None and a.steps < 256 else dev_ms / a.steps
    if many and a.steps < 256:
        # (per vector step, from launches of the timed region's own length: the median of the event-bracketed ones above)
        kern_ms = durs[len(durs) // 2] / a.steps_per_launch_used
    failed_frac = float(env.failed.float().mean().item())
    iters_mean = float(env.peek("PF_ITERS").float().mean().item())
    sweeps_mean = float(env.peek("PF_SWEEPS").float().mean().item())
    n_agents_env, n_bus_env, used_graph = env.n_agents, env.n_bus, bool(graphs)

    # sibling figure with the OTHER solver (north_star names Newton-Raphson; the headline runs the sweep solver whose result
    # Newton's fp64 mismatch test verifies): same step, same inputs, >= 512 launches as 16-step graphs, HIP events
    sibling = None
    if not a.no_sustained:
        other = "newton" if a.solver == "sweep" else "sweep"
        try:
            env2 = VecFlexProvisionEnv({}, a.envs, device=f"cuda:{local_rank}", net=net, series=series,
                                       seed=1234 + 1000 * rank, warm_start=bool(a.warm_start), pf_tol=a.pf_tol,
                                       solver={"sweep": 2, "newton": 0}[other])
            env2.reset()

            def step2(k):
                env2.step(pool[k % ACTION_POOL], fuse_obs=True, auto_reset=True)

            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                step2(0)
            torch.cuda.current_stream().wait_stream(side)
            g2 = torch.cuda.CUDAGraph()
            with graph_capture(g2):
                for j in range(ACTION_POOL):
                    step2(j)
            for _ in range(4):
                g2.replay()
            n2 = 512
            barrier()
            q0, q1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t2 = time.perf_counter()
            q0.record()
            for _ in range(n2 // ACTION_POOL):
                g2.replay()
            q1.record()
            barrier()
            el2 = max_over_ranks(time.perf_counter() - t2)
            sibling = {"solver": other, "steps": n2, "value": a.envs * world * n2 / el2, "unit": "env-steps/s",
                       "device_ms_per_step": q0.elapsed_time(q1) / n2,
                       "pf_newton_iters_mean": float(env2.peek("PF_ITERS").float().mean().item()),
                       "pf_sweeps_mean": float(env2.peek("PF_SWEEPS").float().mean().item()),
                       "solver_failed_frac": float(env2.failed.float().mean().item())}
            del env2, g2
        except Exception as exc:
            print(f"[bench] rank {rank}: solver sibling leg failed: {exc!r}", file=sys.stderr)

    # sibling figure with the OTHER launch form (one flexenv_step launch per step, HIP graphs of 16 <-> flexenv_step_many): same
    # steps, same inputs, same results; 1024 steps, HIP events on the launch stream
    form_sibling = None
    if not a.no_sustained and not a.stacked_obs:
        try:
            env6 = VecFlexProvisionEnv({}, a.envs, device=f"cuda:{local_rank}", net=net, series=series,
                                       seed=1234 + 1000 * rank, warm_start=bool(a.warm_start), pf_tol=a.pf_tol,
                                       solver={"sweep": 2, "newton": 0}[a.solver], sweep_accel=not a.no_sweep_accel)
            env6.reset()
            n6 = 1024
            if many:
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    env6.step(pool[0], obs_rows=True, auto_reset=True)
                torch.cuda.current_stream().wait_stream(side)
                g6 = torch.cuda.CUDAGraph()
                with graph_capture(g6):
                    for j in range(ACTION_POOL):
                        env6.step(pool[j], obs_rows=True, auto_reset=True)

                def run6():
                    for _ in range(n6 // ACTION_POOL):
                        g6.replay()
                form6, spl6 = "one flexenv_step launch per step (HIP graphs of 16 launches): the headline form of rounds 1-4", 1
            else:
                spl6 = max(1, a.steps_per_launch)
                out6 = (torch.empty(spl6, a.envs, dtype=torch.float64, device=dev), torch.empty(spl6, a.envs, dtype=torch.uint8, device=dev),
                        torch.empty(spl6, a.envs, 7, dtype=torch.float64, device=dev), torch.empty(spl6, a.envs, dtype=torch.uint8, device=dev))
                n6 = max(1, n6 // spl6) * spl6

                def run6():
                    for _ in range(n6 // spl6):
                        env6.step_many(pool, steps=spl6, auto_reset=True, out=out6)
                form6 = "flexenv_step_many, %d steps per launch" % spl6
            run6()
            barrier()
            q0, q1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t6 = time.perf_counter()
            q0.record()
            run6()
            q1.record()
            barrier()
            el6 = max_over_ranks(time.perf_counter() - t6)
            ms6 = q0.elapsed_time(q1) / n6
            form_sibling = {"launch_form": form6, "steps_per_launch": spl6, "steps": n6, "value": a.envs * world * n6 / el6,
                            "unit": "env-steps/s", "device_ms_per_step": ms6, "avg_launch_ms": ms6 * spl6,
                            "algorithmic_bytes_per_env_step": B_ALG_CORE,
                            "frac": B_ALG_CORE * a.envs / (ms6 * 1e-3) / 1e9 / HBM_PEAK_GBS}
            del env6
        except Exception as exc:
            print(f"[bench] rank {rank}: launch-form sibling leg failed: {exc!r}", file=sys.stderr)

    # sibling figure with the OTHER observation form (stacked copy per step <-> row push): same step, same inputs
    obs_sibling = None
    if not a.no_sustained:
        try:
            env3 = VecFlexProvisionEnv({}, a.envs, device=f"cuda:{local_rank}", net=net, series=series,
                                       seed=1234 + 1000 * rank, warm_start=bool(a.warm_start), pf_tol=a.pf_tol,
                                       solver={"sweep": 2, "newton": 0}[a.solver])
            env3.reset()
            kw3 = dict(obs_rows=True) if a.stacked_obs else dict(fuse_obs=True)

            def step3(k):
                env3.step(pool[k % ACTION_POOL], auto_reset=True, **kw3)

            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                step3(0)
            torch.cuda.current_stream().wait_stream(side)
            g3 = torch.cuda.CUDAGraph()
            with graph_capture(g3):
                for j in range(ACTION_POOL):
                    step3(j)
            for _ in range(8):
                g3.replay()
            n3 = 1024
            barrier()
            q0, q1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t3 = time.perf_counter()
            q0.record()
            for _ in range(n3 // ACTION_POOL):
                g3.replay()
            q1.record()
            barrier()
            el3 = max_over_ranks(time.perf_counter() - t3)
            ms3 = q0.elapsed_time(q1) / n3
            b3 = B_ALG_CORE if a.stacked_obs else B_ALG_WITH_OBS
            obs_sibling = {"observation": "row push (FLEX_STEP_OBS_ROWS)" if a.stacked_obs else "stacked [n_agents, 6 * history] fp32 copy per step (round-3 form)",
                           "steps": n3, "value": a.envs * world * n3 / el3, "unit": "env-steps/s", "device_ms_per_step": ms3,
                           "algorithmic_bytes_per_env_step": b3, "frac": b3 * a.envs / (ms3 * 1e-3) / 1e9 / HBM_PEAK_GBS}
            del env3, g3
        except Exception as exc:
            print(f"[bench] rank {rank}: observation-form sibling leg failed: {exc!r}", file=sys.stderr)

    # sibling figure at the tolerance BASELINE.json's north_star states for correctness (voltages and rewards within 1e-6 of
    # the CPU reference): the headline iterates to a power mismatch of 1e-12 pu — tighter than the reference's own IPOPT solve
    # (pf.py:101-102, default tolerance 1e-8) — which is a choice, not the contract.  Same step, same inputs, pf_tol = 1e-6;
    # the deviation of its voltages from the headline's after the same steps from the same seed is measured and reported.
    tol_sibling = None
    if not a.no_sustained and a.pf_tol < 1e-6:
        try:
            envs_t = []
            for tol in (1e-6, a.pf_tol):
                e_ = VecFlexProvisionEnv({}, a.envs, device=f"cuda:{local_rank}", net=net, series=series,
                                         seed=4321 + 1000 * rank, warm_start=bool(a.warm_start), pf_tol=tol,
                                         solver={"sweep": 2, "newton": 0}[a.solver])
                e_.reset()
                envs_t.append(e_)
            env4, env5 = envs_t
            for k in range(3 * ACTION_POOL):                      # the same trajectory on both (same seed, same actions)
                env4.step(pool[k % ACTION_POOL], obs_rows=True, auto_reset=True)
                env5.step(pool[k % ACTION_POOL], obs_rows=True, auto_reset=True)
            dv = float((env4.peek("V").double() - env5.peek("V").double()).abs().max().item())
            g4 = torch.cuda.CUDAGraph()
            with graph_capture(g4):
                for j in range(ACTION_POOL):
                    env4.step(pool[j % ACTION_POOL], obs_rows=True, auto_reset=True)
            for _ in range(8):
                g4.replay()
            n4 = 1024
            barrier()
            q0, q1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t4 = time.perf_counter()
            q0.record()
            for _ in range(n4 // ACTION_POOL):
                g4.replay()
            q1.record()
            barrier()
            el4 = max_over_ranks(time.perf_counter() - t4)
            tol_sibling = {"pf_tol": 1e-6, "what": "the tolerance north_star states (1e-6); the headline runs at pf_tol "
                           f"{a.pf_tol:g}", "steps": n4, "value": a.envs * world * n4 / el4, "unit": "env-steps/s",
                           "device_ms_per_step": q0.elapsed_time(q1) / n4,
                           "pf_sweeps_mean": float(env4.peek("PF_SWEEPS").float().mean().item()),
                           "pf_newton_iters_mean": float(env4.peek("PF_ITERS").float().mean().item()),
                           "solver_failed_frac": float(env4.failed.float().mean().item()),
                           "max_abs_dV_vs_headline_tolerance": dv, "after_steps": 3 * ACTION_POOL}
            del env4, env5, g4
        except Exception as exc:
            print(f"[bench] rank {rank}: tolerance sibling leg failed: {exc!r}", file=sys.stderr)

    # the line's headline part is assembled BEFORE the training legs (what they add is filled in afterwards)
    out = None
    if rank == 0:
        total_env_steps = a.envs * world * a.steps
        # algorithmic bytes per env-step of the kernel AS IT RUNS: 1 340 B with get_obs() as a row push (the 240 B
        # of new observation features are the two fp32 mirror rows), 4 220 B when the stacked fp32 observation is materialised
        b_alg = B_ALG_WITH_OBS if a.stacked_obs else B_ALG_CORE
        achieved = b_alg * a.envs / (kern_ms * 1e-3) / 1e9
        roof_extra = roofline_details(a, env, kern_ms)
        out = {
            # BASELINE.json's metric string verbatim; the "+ PF-kernel HBM GB/s" half is `roofline.achieved`
            "metric": "env-steps/sec (33-bus, 4096 envs/GPU) at 1/2/4/8 MI355X + PF-kernel HBM GB/s",
            "value": total_env_steps / elapsed,
            "unit": "env-steps/s",
            "n_gpus": world,
            "world_size": dist_info["world_size"] if dist_info else 1,
            "backend": dist_info["backend"] if dist_info else None,
            "dist": dist_info,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3,
            "value_device_events": total_env_steps / (dev_ms_max * 1e-3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic (stand-in IEEE-33 Baran-Wu network, SURVEY.md App. C; generated series, SURVEY.md §8d)",
            "config": {
                # (the driver keeps the first 120 characters of this string: envs, solver, tolerance and observation form first)
                "workload": ("step()+get_obs() %d envs/GPU, 33-bus PF, %d agents; %s; %s, tol %g; %s obs" % (
                    a.envs, n_agents_env, ("%d steps/launch" % a.steps_per_launch_used) if many else "1 step/launch",
                    "sweeps+fp64 Newton check" if a.solver == "sweep" else "fp64 Newton (tree)", a.pf_tol,
                    "stacked" if a.stacked_obs else "row-push")),
                "workload_detail": ("flex_provision.step()+get_obs() batched, in-launch auto-reset; get_obs(): %s; solver: %s, "
                                    "inf-norm power mismatch < %g pu") % (
                                 "stacked [5, 144] fp32 copy per step" if a.stacked_obs else
                                 "row push — the step appends its [5, 6] feature row to the env's observation history (mirror ring); "
                                 "consumers read the stacked [5, 144] window in place (policy kernels) or via flexenv_obs_view",
                                 "backward/forward sweeps (fp64 anchor sweeps, fp32 increment sweeps between them, two-sweep "
                                 "extrapolation %s) + fp64 Newton verification of the Ybus mismatch (%.3g Newton steps, %.2f sweeps "
                                 "per solve)" % ("off" if a.no_sweep_accel else "on", iters_mean, sweeps_mean)
                                 if a.solver == "sweep" else
                                 "fp64 Newton-Raphson on the Ybus, tree-structured elimination (%.2f Newton steps per solve)" % iters_mean,
                                 a.pf_tol),
                "pf_tol": a.pf_tol,
                "arith": ("fp64 verify / fp32 increments" if a.solver == "sweep" else "fp64"),
                "obs_form": "stacked" if a.stacked_obs else "row_push",
                "sweep_accel": bool(a.solver == "sweep" and not a.no_sweep_accel),
                "envs_per_gpu": a.envs, "n_agents": n_agents_env, "n_bus": n_bus_env,
                "launch_form": ("many: flexenv_step_many — the K steps as launches of up to %d steps (the timed region: %s), every wavefront "
                                "walking its own environments through the sequence (run_env.py:78-92 vectorised); per step the same loads, "
                                "stores and arithmetic as one flexenv_step launch, results bit-identical (tests/test_step_many_gpu.py); "
                                "`single_launch_sibling` is the one-launch-per-step form timed in this run" % (
                                    max(1, a.steps_per_launch),
                                    " + ".join(["%d x %d" % (a.steps // max(1, a.steps_per_launch), max(1, a.steps_per_launch))] * (a.steps >= max(1, a.steps_per_launch))
                                               + ["1 x %d" % (a.steps % max(1, a.steps_per_launch))] * (a.steps % max(1, a.steps_per_launch) > 0))))
                               if many else "single: one flexenv_step launch per step",
                "steps_per_launch": a.steps_per_launch_used,
                "warm_start": bool(a.warm_start), "launches_per_step": (1.0 / a.steps_per_launch_used), "hip_graph": used_graph, "hip_graph_uploaded": used_graph,
                "timing": ("barrier + synchronize | K steps | barrier + synchronize, clock; MAX over ranks (value_device_events: MAX "
                           "over ranks of the HIP-event time of each rank's K steps)" if distributed else "synchronize | K steps | synchronize"),
                "timed_after": ("the `sustained` leg (%d steps), then W warm-up steps" % sustained["steps"]) if sustained else "W warm-up steps",
                "device_ms_per_step": dev_ms / a.steps, "solver": ("sweep (mixed fp64/fp32 increments) + fp64 Newton verification" if a.solver == "sweep" else "newton (fp64, tree elimination)"),
                "pf_newton_iters_mean": iters_mean, "pf_sweeps_mean": sweeps_mean,
                "pf_sweeps_counts": "sweeps executed by the environment's wavefront (two environments per wavefront run until both have passed)",
                "solver_failed_frac": failed_frac,
            },
            "roofline": dict({
                "bound": "hbm",
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "algorithmic_bytes_per_env_step": b_alg,
                # the SAME launch time priced at the other byte convention of SURVEY.md 8(d), so that the figure stays
                # comparable across rounds (rounds 1-3 materialised the stacked observation and quoted 4 220 B: 0.154 / 0.159 /
                # 0.164); `stacked_sibling` below is the kernel that really moves those bytes, timed in this run
                "frac_at_4220_bytes": B_ALG_WITH_OBS * a.envs / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "frac_at_1340_bytes": B_ALG_CORE * a.envs / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                # one launch = `steps_per_launch` vector steps of `envs_per_gpu` environments: algorithmic bytes per launch =
                # algorithmic_bytes_per_env_step x envs x steps_per_launch, over avg_launch_ms
                "steps_per_launch": a.steps_per_launch_used, "device_ms_per_vector_step": kern_ms,
                "avg_launch_ms": kern_ms * a.steps_per_launch_used, "bracketed_launch_ms": durs[len(durs) // 2],
            }, **roof_extra),
            "single_launch_sibling" if many else "many_steps_sibling": form_sibling,
            "stacked_sibling" if not a.stacked_obs else "rows_sibling": obs_sibling,
            "sustained": sustained,
            "solver_sibling": sibling,
            "tolerance_sibling": tol_sibling,
            "train": None,
            "train_kernel_shares": None,
            "learner_rooflines": None,
        }

    # training legs LAST: the headline (which the scaling curve is computed from) is already measured if a leg fails; every
    # rank takes part — with N > 1 the gradient bucket goes through RCCL — and a failure is recorded, not fatal
    train = None
    train_failed = False
    emitted = []
    deadline = None
    if distributed and not a.no_train and a.train_deadline > 0:
        # N > 1 only.  A collective that cannot complete inside a training leg (a rank that died, a transport fault) ends in the
        # process group's own abort after its 180 s timeout — which would take rank 0 down BEFORE it printed the line whose headline
        # part (env-steps/s over all ranks, already measured above) the scaling curve is computed from.  So: if the legs have not
        # finished after --train-deadline seconds, every rank stops waiting; rank 0 prints the ONE line with what the legs recorded
        # so far plus the reason, and the processes leave without the group's shutdown handshake (it could not complete either).
        import threading

        def _give_up():
            if emitted:
                return
            emitted.append("deadline")
            print(f"[bench] rank {rank}: training legs not finished after {a.train_deadline} s: giving up on them", file=sys.stderr, flush=True)
            if rank == 0:
                out.update({"train": list(train or []) + [{"error": "training legs not finished after %d s (--train-deadline); "
                                                           "the headline above was measured before them" % a.train_deadline}]})
                print(json.dumps(out), flush=True)
            os._exit(0)

        deadline = threading.Timer(a.train_deadline, _give_up)
        deadline.daemon = True
        deadline.start()
    if not a.no_train:
        # (alg, agents, envs per GPU, batch divisor): every configuration at the trainer's default batch (envs / 4 x 32:
        # 1.47 samples per transition) AND at the reference's own sample reuse (envs x 32: 5.87)
        if distributed:
            legs = [("maddpg", 5, N_ENVS, 4, False), ("safemaddpg", 5, 2 * N_ENVS, 4, False)]
        else:
            # the last leg: config 4 with a policy that CAN act on the safety layer — labelled, no parity claim
            legs = [("maddpg", 5, N_ENVS, 4, False), ("maddpg", 5, N_ENVS, 1, False), ("maddpg", 3, N_ENVS, 4, False),
                    ("maddpg", 3, N_ENVS, 1, False), ("safemaddpg", 5, 2 * N_ENVS, 4, False), ("safemaddpg", 5, 2 * N_ENVS, 1, False),
                    ("safemaddpg", 5, 2 * N_ENVS, 4, True)]
        train = []
        for alg, n_ag, n_env, div, intended in legs:
            try:
                train.append(train_leg(alg, n_ag, n_env, max(2, a.train_episodes), rank, local_rank, world, barrier,
                                       max_over_ranks, batch_div=div, intended=intended))
            except Exception as exc:                  # a failed leg must not cost the headline line
                print(f"[bench] rank {rank}: training leg {alg}/{n_ag}/{n_env}/{div} failed: {exc!r}", file=sys.stderr)
                train.append({"alg": alg, "n_agents": n_ag, "envs_per_gpu": n_env, "n_gpus": world, "batch_div": div,
                              "error": repr(exc)[:300]})
                train_failed = True
                break

    if deadline is not None:
        deadline.cancel()
        if emitted:                                       # (the timer fired while the last leg was returning: it owns the exit)
            time.sleep(3600)
        emitted.append("legs finished")                   # (a timer already past its cancel point finds this and returns)
    learner = None
    if rank == 0 and not distributed and not a.no_train:
        try:
            learner = learner_rooflines()
        except Exception as exc:
            print(f"[bench] learner rooflines failed: {exc!r}", file=sys.stderr)

    if rank == 0:
        out.update({"train": train, "train_kernel_shares": shares, "learner_rooflines": learner})
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(net, series, a.cpu_seconds, envs=a.envs)
            cb = out["cpu_baseline"]
            # the ratio is reported for orientation only (the roofline fraction is the figure of merit); like for like in the
            # observation form too: the CPU legs write the stacked copy, so `stacked` divides the stacked sibling's rate
            cb["gpu_over_cpu"] = {"headline_row_push": out["value"] / cb["value"],
                                  "stacked": (obs_sibling["value"] / cb["value"]) if obs_sibling and not a.stacked_obs else None}
        print(json.dumps(out), flush=True)
    if distributed:
        try:
            dist.barrier()
            dist.destroy_process_group()
        except Exception as exc:                       # a rank that failed its training leg is not waited for twice
            print(f"[bench] rank {rank}: shutdown: {exc!r}", file=sys.stderr)
    if a.strict and train is not None:
        bad = [t for t in train if "error" in t or not t.get("rollout_graph") or t.get("graphed_updates") != ["policy", "value"]]
        if train_failed or bad:
            print(f"[bench] --strict: {len(bad)} training leg(s) failed or ran without their graphs", file=sys.stderr)
            sys.exit(3)


if __name__ == "__main__":
    main()
