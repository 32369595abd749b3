#!/usr/bin/env python3
"""bench.py — env-steps/s of the batched flexibility-provision step on MI355X.

A "step" is ONE vector step of the hot path over one batch of synthetic input: for every one of the
4096 environments of this GPU, `flexenv_step` (action parse -> 33-bus AC power flow -> ESS update ->
reward, with the `get_obs()` that always follows it fused in, and the restart of the environments that just
terminated) — one kernel launch.  Inputs (series table, action pool) are resident in HBM before the
timed region starts; nothing crosses PCIe inside it.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Environments are independent, so ranks share nothing on the data path (weak scaling, no collective);
the only collectives are the contract's barrier and the max-over-ranks of the elapsed time.
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
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--envs", type=int, default=N_ENVS, help="envs per GPU (the metric is quoted at 4096)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=1.5, help="stepping time of the CPU baseline sample (x cores = CPU work)")
    ap.add_argument("--warm-start", type=int, default=1)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL; gloo only to "
                    "rehearse the multi-rank control flow on a one-GPU box together with FLEX_BENCH_ONE_DEVICE=1)")
    ap.add_argument("--pf-tol", type=float, default=1e-12, help="power-flow convergence threshold (inf-norm power "
                    "mismatch, pu); 1e-12 is the headline setting, the parity bar is 1e-6 on voltages and rewards")
    ap.add_argument("--no-graph", action="store_true", help="launch every step eagerly instead of replaying a HIP graph")
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


def cpu_baseline(net, series, seconds):
    """The oracle's C restatement timed on this box's host cores (rank 0, N=1 only)."""
    import numpy as np
    from oracle import c_oracle
    cores = usable_cores()
    os.environ["OMP_NUM_THREADS"] = str(cores)
    n = 64 * cores
    rng = np.random.default_rng(1234)
    env = c_oracle.COracleEnv(net, series.table, n)
    acts = rng.uniform(0.5, 1.0, (8, n, 5, 4))

    def fresh_episodes():
        day = rng.integers(0, series.n_start_days(96), n)
        start = rng.integers(0, 4, n) + rng.integers(0, 24, n) * 4 + day * 96
        env.reset(start, rng.uniform(0.01125, 0.01375, (n, 5)), rng.uniform(0, 1, (n, 20)))

    fresh_episodes()
    for k in range(4):                                   # page in, spin up the OpenMP team
        env.step(acts[k % 8])
    # bounded sample: whole 90-step episodes (no resets inside the timed steps) until `seconds` of stepping
    busy, done_steps = 0.0, 0
    while busy < seconds:
        fresh_episodes()
        t0 = time.perf_counter()
        for k in range(90):
            env.step(acts[k % 8])
        busy += time.perf_counter() - t0
        done_steps += 90
    return {"value": n * done_steps / busy, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": f"{n} envs x {done_steps} steps of step()+get_obs() (C restatement oracle/flexenv_oracle.c, dense polar "
                      f"NR, OpenMP over envs, {cores} threads), {busy:.1f} s = {busy * cores:.0f} core-seconds"}


def main():
    a = parse_args()
    import torch
    import safe_marl_amd  # noqa: F401
    from safe_marl_amd.network import create_network
    from safe_marl_amd.series import make_synthetic_series
    from safe_marl_amd.flex_env import VecFlexProvisionEnv

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    if os.environ.get("FLEX_BENCH_ONE_DEVICE") == "1":       # rehearsal only: every rank on cuda:0
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(a.backend)
    coll_dev = dev if a.backend == "nccl" else torch.device("cpu")

    net = create_network()
    series = make_synthetic_series(net)                      # 1096 days x 96 rows x 72 cols fp64 (60.6 MB)
    env = VecFlexProvisionEnv({}, a.envs, device=f"cuda:{local_rank}", net=net, series=series,
                              seed=1234 + 1000 * rank, warm_start=bool(a.warm_start), pf_tol=a.pf_tol,
                              solver={"sweep": 2, "newton": 0}[a.solver])
    gen = torch.Generator(device=dev)
    gen.manual_seed(99 + rank)
    # the range the reference's translate_action actually delivers (util.py:125-128), float32 like util.py:184
    pool = (0.5 + 0.5 * torch.rand(ACTION_POOL, a.envs, env.n_agents, 4, device=dev, generator=gen)).float()
    env.reset()

    def one_step(k):
        # ONE launch: step + get_obs; envs that terminate restart inside the same launch (FLEX_STEP_AUTORESET)
        env.step(pool[k % ACTION_POOL], fuse_obs=True, auto_reset=True)

    # The loop is launch-issue sensitive (9 us of Python + ctypes per launch against a 16 us kernel), so ACTION_POOL
    # consecutive steps are captured once as a HIP graph and replayed; the steps that do not fill a graph run eagerly.
    # Work per step is identical either way (one flexenv_step launch); `--no-graph` keeps everything eager.
    graph = None
    if not a.no_graph:
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                one_step(0)
            torch.cuda.current_stream().wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for j in range(ACTION_POOL):
                    one_step(j)
            graph = g
        except Exception as exc:                      # capture not available: eager launches
            print(f"[bench] HIP graph capture failed ({exc}); eager launches", file=sys.stderr)
            graph = None

    def run_steps(count, first):
        done_steps = 0
        if graph is not None:
            while count - done_steps >= ACTION_POOL:
                graph.replay()
                done_steps += ACTION_POOL
        for k in range(done_steps, count):
            one_step(first + k)

    run_steps(a.warmup, 0)

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    run_steps(a.steps, a.warmup)                          # exactly K steps
    ev1.record()
    barrier()
    elapsed = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # roofline leg.  The timed region above is K back-to-back launches of ONE kernel (flex_step_kernel) on one
    # stream, bracketed by the HIP events ev0/ev1 on that stream: dev_ms / K is its average launch duration
    # (rocprofv3 --kernel-trace --stats of the same command agrees: profiles/).  A second pass brackets every
    # launch with its own event pair; that figure carries ~2 us of event overhead per launch and is reported
    # as `bracketed_launch_ms` only.
    n_ev = min(a.steps, 400)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_ev)]
    torch.cuda.synchronize()
    for k, (s, e) in enumerate(evs):
        s.record()
        env.step(pool[k % ACTION_POOL], fuse_obs=True, auto_reset=True)
        e.record()
    torch.cuda.synchronize()
    durs = sorted(s.elapsed_time(e) for s, e in evs)
    kern_ms = dev_ms / a.steps
    failed_frac = float(env.failed.float().mean().item())
    iters_mean = float(env.peek("PF_ITERS").float().mean().item())
    sweeps_mean = float(env.peek("PF_SWEEPS").float().mean().item())

    if rank == 0:
        total_env_steps = a.envs * world * a.steps
        achieved = B_ALG_WITH_OBS * a.envs / (kern_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                t = json.load(open(tpath))
                if int(t.get("envs_per_launch", -1)) == a.envs:      # PMC figure collected at this batch size only
                    traffic = t.get("flex_step_kernel_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            # BASELINE.json's metric string verbatim; the "+ PF-kernel HBM GB/s" half is `roofline.achieved`
            "metric": "env-steps/sec (33-bus, 4096 envs/GPU) at 1/2/4/8 MI355X + PF-kernel HBM GB/s",
            "value": total_env_steps / elapsed,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic (stand-in IEEE-33 Baran-Wu network, SURVEY.md App. C; generated series, SURVEY.md §8d)",
            "config": {
                "workload": "flex_provision.step()+get_obs() batched, 4096 envs/GPU, 33-bus AC power flow (fp64 NR, tol %g), " % a.pf_tol +
                            "5 agents, in-launch auto-reset",
                "envs_per_gpu": a.envs, "n_agents": env.n_agents, "n_bus": env.n_bus,
                "warm_start": bool(a.warm_start), "launches_per_step": 1, "hip_graph": graph is not None,
                "device_ms_per_step": dev_ms / a.steps, "solver": a.solver, "pf_newton_iters_mean": iters_mean, "pf_sweeps_mean": sweeps_mean,
                "solver_failed_frac": failed_frac,
            },
            "roofline": {
                "bound": "hbm", "kernel": "flex_step_kernel<2, float, float, 5>",
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "algorithmic_bytes_per_env_step": B_ALG_WITH_OBS, "algorithmic_bytes_per_env_step_no_obs": B_ALG_CORE,
                "avg_launch_ms": kern_ms, "bracketed_launch_ms": durs[len(durs) // 2],
                "note": "latency/issue-bound fp64 kernel: ~4 KB per env-step cannot approach HBM peak (SURVEY.md §8d)",
            },
        }
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(net, series, a.cpu_seconds)
        print(json.dumps(out), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
