"""Rollout step time, graph-replayed, with and without the fused burst launch (flexenv_rollout_burst): us per vector step
of run(m) over whole episodes.  usage: python tools/burst_bench.py [n_envs] [agents]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))


def main():
    n_envs = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    agents = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    short = len(sys.argv) > 3 and sys.argv[3] == "short"           # (a counter pass: few launches, all of 16 steps)
    from test_rollout_gpu import _trainer
    from safe_marl_amd.learner import RolloutGraph
    for flag in ("0", "1"):
        os.environ["FLEX_ROLLOUT_BURST"] = flag
        tr = _trainer(n_envs, None if agents == 5 else [3, 17, 28][:agents])
        rg = RolloutGraph(tr.behaviour_net, tr.env, tr.replay_buffer)
        rg.start_episode(tr.env.reset())
        rg.capture()
        for m in ((16,) if short else (60, 36, 16, 96)):
            rg.run(m)                                       # warm
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 3 if short else 20
            t0 = time.perf_counter()
            e0.record()
            for _ in range(reps):
                rg.run(m)
            e1.record()
            torch.cuda.synchronize()
            wall = time.perf_counter() - t0
            print(f"envs {n_envs} agents {agents} burst {flag} m {m}: {e0.elapsed_time(e1) * 1e3 / (reps * m):.2f} us/step (events), "
                  f"{wall * 1e6 / (reps * m):.2f} us/step (wall), fused={rg.fused_burst}", flush=True)
        del rg, tr
        import gc
        gc.collect()


if __name__ == "__main__":
    main()
