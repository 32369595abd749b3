#!/usr/bin/env python3
"""The flow of the reference's run_env.py (:54-160) on the device, for N environments at once: reset, then an episode of
sampled actions — `np.random.normal(0, 0.5, n_actions)` per agent (run_env.py:82-85) — stepped open-loop with the reward
recorded per step (run_env.py:92-96), and the wall time of the episode printed (run_env.py:61,158-160).

The whole episode is ONE launch (`VecFlexProvisionEnv.step_many`, include/flexenv.h: flexenv_step_many): every wavefront
walks its own environments through the action sequence.  `--per-step` runs the same episode as one `step()` launch per
step instead; both print the same returns (the results are bit-identical)."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--episodes", type=int, default=1)
    ap.add_argument("--per-step", action="store_true", help="one launch per step (FlexibilityProvisionEnv.step's form)")
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    import torch
    import safe_marl_amd  # noqa: F401
    from safe_marl_amd.flex_env import DEFAULT_ENV_ARGS, VecFlexProvisionEnv
    from safe_marl_amd.network import create_network
    from safe_marl_amd.series import make_synthetic_series

    cfg = DEFAULT_ENV_ARGS
    net = create_network(cfg)
    series = make_synthetic_series(net)
    env = VecFlexProvisionEnv({}, a.envs, device="cuda:0", net=net, series=series, seed=1234 + a.seed, warm_start=True)
    steps = int(cfg["episode_limit"])                                  # max_steps of run_env.py:57,78
    gen = torch.Generator(device="cuda").manual_seed(a.seed)
    print(f"Number of agents: {env.n_agents}; observation size: {env.get_obs_size()}; state size: {env.get_state_size()}")
    for e in range(a.episodes):
        env.reset()
        # run_env.py:82-85: N(0, 0.5) per action; float32 like the policy's output (util.py:184)
        actions = 0.5 * torch.randn(steps, a.envs, env.n_agents, 4, device="cuda", generator=gen)
        torch.cuda.synchronize()
        t0 = time.time()
        if a.per_step:
            rewards = torch.empty(steps, a.envs, dtype=torch.float64, device="cuda")
            failed = torch.zeros(a.envs, dtype=torch.int64, device="cuda")
            for t in range(steps):
                r, _, _ = env.step(actions[t], obs_rows=True)
                rewards[t] = r
                failed += env.failed
        else:
            rewards, dones, infos, faileds = env.step_many(actions)
            failed = faileds.sum(0)
        torch.cuda.synchronize()
        dt = time.time() - t0
        ret = rewards.sum(0)
        print(f"episode {e}: {a.envs} envs x {steps} steps in {dt * 1e3:.2f} ms ({a.envs * steps / dt / 1e6:.1f} M env-steps/s); "
              f"return mean {ret.mean().item():+.4f}, min {ret.min().item():+.4f}, max {ret.max().item():+.4f}; "
              f"unsolved power flows: {int(failed.sum().item())}; min |V| at the end {env.peek('V').min().item():.4f} pu")


if __name__ == "__main__":
    main()
