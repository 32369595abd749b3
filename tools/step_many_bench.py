"""A/B on one box: the headline loop (one flexenv_step launch per vector step, replayed as HIP graphs of 16 launches) against
flexenv_step_many (the same steps in ONE launch), with and without the register carry, at several launch lengths.
Same workload as bench.py's headline: 4096 envs, warm-started sweep solver, row push, in-launch auto-reset, actions from a
pool of 16 slabs in [0.5, 1).  Times with HIP events on the current stream; prints one line per form.

    python tools/step_many_bench.py [--envs 4096] [--steps 1520] [--reps 5]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=1520)
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    import torch
    import safe_marl_amd  # noqa: F401
    from safe_marl_amd.network import create_network
    from safe_marl_amd.series import make_synthetic_series
    from safe_marl_amd.flex_env import VecFlexProvisionEnv
    from safe_marl_amd.util import graph_capture

    net = create_network()
    series = make_synthetic_series(net)
    pool_n = 16
    gen = torch.Generator(device="cuda").manual_seed(99)
    pool = (0.5 + 0.5 * torch.rand(pool_n, a.envs, 5, 4, device="cuda", generator=gen)).float()

    def fresh():
        env = VecFlexProvisionEnv({}, a.envs, device="cuda:0", net=net, series=series, seed=1234, warm_start=True)
        env.reset()
        return env

    def timed(fn, reps):
        fn()
        torch.cuda.synchronize()
        best = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            best.append(e0.elapsed_time(e1) * 1e3)
        best.sort()
        return best[len(best) // 2], best[0], best[-1]

    # --- one launch per step, graphs of 16 ---------------------------------------------------------------------------
    env = fresh()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        env.step(pool[0], obs_rows=True, auto_reset=True)
    torch.cuda.current_stream().wait_stream(side)
    with graph_capture(g):
        for j in range(pool_n):
            env.step(pool[j], obs_rows=True, auto_reset=True)
    blocks = a.steps // pool_n

    def single():
        for _ in range(blocks):
            g.replay()
    med, lo, hi = timed(single, a.reps)
    n_steps = blocks * pool_n
    print(f"single-step launches (graphs of 16)    : {med / n_steps:7.3f} us per vector step (min {lo / n_steps:.3f}, max {hi / n_steps:.3f})"
          f"  -> {a.envs * n_steps / med:8.1f} M env-steps/s")

    # --- many steps per launch -----------------------------------------------------------------------------------------
    for per_launch in (16, 20, 95, n_steps):
        for carry in (True, False):
            env = fresh()
            out = (torch.empty(per_launch, a.envs, dtype=torch.float64, device="cuda"),
                   torch.empty(per_launch, a.envs, dtype=torch.uint8, device="cuda"),
                   torch.empty(per_launch, a.envs, 7, dtype=torch.float64, device="cuda"),
                   torch.empty(per_launch, a.envs, dtype=torch.uint8, device="cuda"))
            launches = max(1, n_steps // per_launch)

            def many():
                for _ in range(launches):
                    env.step_many(pool, steps=per_launch, auto_reset=True, out=out, carry=carry)
            med, lo, hi = timed(many, a.reps)
            tot = launches * per_launch
            print(f"step_many {per_launch:5d} steps/launch carry={int(carry)}  : {med / tot:7.3f} us per vector step (min {lo / tot:.3f}, max {hi / tot:.3f})"
                  f"  -> {a.envs * tot / med:8.1f} M env-steps/s")


if __name__ == "__main__":
    main()
