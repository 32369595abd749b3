"""How the host learns that the driver's 20-step region is done: torch.cuda.synchronize() alone against polling the closing
event first (ev.query() in a loop, then the synchronize).  Same launch as bench.py's timed region (one prepared flexenv_step_many
launch of 20 steps, 4096 envs); interleaved repetitions, wall clock between the opening synchronize and the return of the closing
one.  python tools/sync_wait_probe.py [--reps 200]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=200)
    ap.add_argument("--steps", type=int, default=20)
    a = ap.parse_args()
    import torch
    import safe_marl_amd  # noqa: F401
    from safe_marl_amd.network import create_network
    from safe_marl_amd.series import make_synthetic_series
    from safe_marl_amd.flex_env import VecFlexProvisionEnv
    net = create_network()
    series = make_synthetic_series(net)
    env = VecFlexProvisionEnv({}, 4096, device="cuda:0", net=net, series=series, seed=1234, warm_start=True)
    env.reset()
    gen = torch.Generator(device="cuda").manual_seed(99)
    pool = (0.5 + 0.5 * torch.rand(16, 4096, 5, 4, device="cuda", generator=gen)).float()
    launch = env.step_many_prepared(pool, steps=a.steps, auto_reset=True)[0]
    big = env.step_many_prepared(pool, steps=256, auto_reset=True)[0]
    for _ in range(8):
        big()                                   # operating clocks
    torch.cuda.synchronize()
    res = {"sync": [], "poll": [], "dev": []}
    for r in range(a.reps):
        for mode in ("sync", "poll"):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            big()
            torch.cuda.synchronize()
            e0.record()
            t0 = time.perf_counter()
            launch()
            e1.record()
            if mode == "poll":
                while not e1.query():
                    pass
            torch.cuda.synchronize()
            res[mode].append((time.perf_counter() - t0) * 1e6)
            res["dev"].append(e0.elapsed_time(e1) * 1e3)
    for k, v in res.items():
        v.sort()
        print(f"{k:5s}: median {v[len(v) // 2]:8.2f} us   p10 {v[len(v) // 10]:8.2f}   p90 {v[9 * len(v) // 10]:8.2f}   ({len(v)} samples)")


if __name__ == "__main__":
    main()
