#!/usr/bin/env python3
"""Long-run behaviour of the MADDPG update: policy loss, value-loss maxima, BatchNorm running statistics and
the test reward on FIXED evaluation episodes every --every episodes, for

    --mode vec   the vectorised trainer (VecFlexProvisionEnv, --envs, batch_scale = envs / --div)
    --mode n1    the N = 1 reference-cadence control (FlexibilityProvisionEnv, batch 32, model.py:198-267 to the letter):
                 the same number of update events per episode, i.e. the same number of gradient steps

and next to every record the value the critic SHOULD report for the evaluated policy under the normalised reward of
model.py:321-322 — Q_pred = mean over an episode's positions t of sum_{j < 95 - t} gamma^j (r_pi - mu) / sigma, with r_pi the
policy's mean test reward per step and (mu, sigma) the reward BatchNorm's running statistics — so that "the policy loss runs
to -37" can be read against what -mean Q(s, pi(s)) ought to be.

Exit code 2 (a failed soak) when, at the end: the test reward lies below --floor (default 0.95) of its running best, or
|policy loss| exceeds --qmax (default 100: twice what the N = 1 reference-cadence control reaches in 1500 episodes,
profiles/r04_qdrift_n1.json: 47.5; the discounted horizon is 61 steps, i.e. 100 is 1.6 sigma of reward per step sustained), or
any statistic is non-finite.  (Q_pred is reported, not gated on: the reference's DDPG update over-estimates it by two orders of
magnitude on BOTH paths)

    python tools/q_drift.py --mode vec --envs 4096 --div 4 --episodes 1500 --out gpurun_out/r04_qdrift_vec4.json
"""
import argparse
import contextlib
import io
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "examples"))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", choices=["vec", "n1"], default="vec")
    ap.add_argument("--alg", default="maddpg")
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--div", type=int, default=4)
    ap.add_argument("--episodes", type=int, default=1500)
    ap.add_argument("--every", type=int, default=50)
    ap.add_argument("--eval-envs", type=int, default=1024)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--floor", type=float, default=0.95)
    ap.add_argument("--qmax", type=float, default=100.0)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()

    import numpy as np
    import torch
    import safe_marl_amd  # noqa: F401
    from train_maddpg import DEFAULT_ALG_ARGS
    from learning_curve import eval_spec
    from safe_marl_amd import learner
    from safe_marl_amd.flex_env import FlexibilityProvisionEnv, VecFlexProvisionEnv
    from safe_marl_amd.network import create_network
    from safe_marl_amd.series import make_synthetic_series
    from safe_marl_amd.trainer import PGTrainer
    from safe_marl_amd.util import convert

    net = create_network({})
    series = make_synthetic_series(net, n_days=365)
    n_agents = len(net["buildings"])
    spec = eval_spec(series, a.eval_envs, n_agents)
    ev = VecFlexProvisionEnv({}, a.eval_envs, net=net, series=series, seed=77, warm_start=True)
    d = dict(DEFAULT_ALG_ARGS)
    d.update(alg=a.alg, agent_num=n_agents, obs_size=144, state_size=110, action_dim=4, v_min=0.9, v_max=1.1)
    args = convert(d)
    torch.manual_seed(a.seed)
    np.random.seed(a.seed)
    cls = {"maddpg": learner.MADDPG, "safemaddpg": learner.SAFEMADDPG}[a.alg]
    if a.mode == "vec":
        env = VecFlexProvisionEnv({}, a.envs, net=net, series=series, seed=1234, warm_start=True)
        tr = PGTrainer(args, cls, env, None, batch_scale=max(1, a.envs // a.div), replay_capacity=a.envs * 96 * 2)
    else:
        env = FlexibilityProvisionEnv({}, net=net, series=series, warm_start=True)
        tr = PGTrainer(args, cls, env, None)
    gamma, T = float(args.gamma), 95
    # mean over positions t = 0..T-1 of the discounted number of remaining steps
    horizon = float(np.mean([(1 - gamma ** (T - t)) / (1 - gamma) for t in range(T)]))
    net_ = tr.behaviour_net
    rec, best = [], -1e9
    win = {"ploss": [], "vloss": [], "reward": []}
    t0 = time.perf_counter()
    for ep in range(a.episodes + 1):
        if ep % a.every == 0 or ep == a.episodes:
            e = net_.evaluate_on(ev, spec)
            test = float(e["mean_test_reward"])
            best = max(best, test)
            mu = float(net_.batchnorm.running_mean.float().mean().item())
            sig = float(net_.batchnorm.running_var.float().mean().sqrt().item())
            r = {"episode": ep, "test_reward": test, "best_test_reward": best, "bn_running_mean": mu, "bn_running_std": sig,
                 "q_pred": (test - mu) / max(sig, 1e-12) * horizon,
                 "policy_loss_mean": float(np.mean(win["ploss"])) if win["ploss"] else None,
                 "value_loss_mean": float(np.mean(win["vloss"])) if win["vloss"] else None,
                 "value_loss_max": float(np.max(win["vloss"])) if win["vloss"] else None,
                 "train_reward_mean": float(np.mean(win["reward"])) if win["reward"] else None,
                 "seconds": time.perf_counter() - t0}
            rec.append(r)
            print(json.dumps(r), flush=True)
            win = {"ploss": [], "vloss": [], "reward": []}
        if ep == a.episodes:
            break
        stat = {}
        with contextlib.redirect_stdout(io.StringIO()):           # (the N = 1 env prints a line per episode, env:351)
            net_.train_process(stat, tr)
        for k, v in list(stat.items()):
            if isinstance(v, torch.Tensor):
                stat[k] = float(v.item())
        bad = [k for k, v in stat.items() if not np.isfinite(v)]
        if bad:
            print("non-finite", ep, bad)
            sys.exit(2)
        if "mean_train_policy_loss" in stat:
            win["ploss"].append(stat["mean_train_policy_loss"])
            win["vloss"].append(stat["mean_train_value_loss"])
        win["reward"].append(stat["mean_train_reward"])
    out = {"mode": a.mode, "alg": a.alg, "envs": a.envs if a.mode == "vec" else 1,
           "batch": tr.effective_batch_size(), "episodes": a.episodes, "gamma": gamma, "mean_discounted_horizon": horizon,
           "flags": {k: v for k, v in os.environ.items() if k.startswith("FLEX_")}, "records": rec}
    last = rec[-1]
    fails = []
    if last["test_reward"] < a.floor * best:
        fails.append(f"test reward {last['test_reward']:.5f} below {a.floor} x its running best {best:.5f}")
    if last["policy_loss_mean"] is not None and abs(last["policy_loss_mean"]) > a.qmax:
        fails.append(f"|policy loss| {abs(last['policy_loss_mean']):.2f} above {a.qmax} (Q_pred {last['q_pred']:.2f})")
    out["fails"] = fails
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        json.dump(out, open(a.out, "w"), indent=1)
    print("q_drift", "FAILED: " + "; ".join(fails) if fails else "ok", flush=True)
    sys.exit(2 if fails else 0)


if __name__ == "__main__":
    main()
