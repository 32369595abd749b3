#!/usr/bin/env python3
"""Generates tests/golden/loop_golden.npz: the reference's WHOLE training loop, executed — utils/trainer.PGTrainer +
madrl/models/maddpg.MADDPG + utils/replay_buffer.TransReplayBuffer + madrl/environments/flex_provision/
flexibility_provision_env.FlexibilityProvisionEnv, three episodes of Model.train_process (model.py:198-267: 285 env steps,
update events of ten value + one policy sub-update at steps 60 / 120 / 180 / 240, soft target updates at 120 / 240) on CPU
from fixed seeds — with the three substitutions of make_env_golden.py (power_flow_solver := oracle/pf_oracle Newton-Raphson,
read_excel := the package's xlsx reader on stand-in workbooks, synthetic CSVs) and nothing else changed.

    cd /root/reference && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/make_loop_golden.py

Then Model.evaluation (model.py:269-306): ten test-mode episodes.  Recorded: the initial state_dict (behaviour + target),
every step's env action / reward / done (training and evaluation), every episode's statistics, the evaluation's mean_test_*
dictionary, the record of utils/tester.PGTester.run(3, 7, 1) (tester.py:23-33 keys), the final state_dict.  The GPU test drives the PRODUCT's N = 1 path (drop-in env on the HIP kernels, the package's trainer /
learner / replay on CPU tensors so that torch's CPU generator draws the reference's exploration noise) from the same seeds
and must land on the same trajectory and the same weights.
"""
import contextlib
import io
import os
import sys

import numpy as np
import torch as th

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_env_golden as M  # noqa: E402  (chdir to /root/reference, sys.path, pyomo placeholders, TMP inputs)
import make_learner_golden as L  # noqa: E402  (load_args: default.yaml + maddpg.yaml + env args, as train_agent.py merges them)

OUT = os.environ.get("GOLDEN_OUT", HERE)


def main():
    import pandas as pd
    import yaml
    M.write_inputs()
    pd.read_excel = M.fake_read_excel
    with open("madrl/args/env_args/flex_provision.yaml") as f:
        env_args = yaml.safe_load(f)["env_args"]
    from madrl.environments.flex_provision import flexibility_provision_env as E
    from oracle import pf_oracle
    E.power_flow_solver = lambda *a: pf_oracle.power_flow_solver(*a, env_config=env_args)
    from utils.trainer import PGTrainer
    from utils.util import convert
    from madrl.models.maddpg import MADDPG

    kw = dict(env_args)
    kw.update(data_path=M.TMP, seed=11)
    with contextlib.redirect_stdout(io.StringIO()):
        env = E.FlexibilityProvisionEnv(kw)                     # np.random.seed(11); reset()
    argd = L.load_args()
    argd.update(agent_num=env.get_num_of_agents(), obs_size=env.get_obs_size(), state_size=env.get_state_size(),
                action_dim=env.get_total_actions(), cuda=False)
    args = convert(argd)
    th.manual_seed(2024)
    trainer = PGTrainer(args, MADDPG, env, None)
    g = {}
    for k, v in trainer.behaviour_net.state_dict().items():
        g["init." + k] = v.detach().cpu().numpy().copy()
    # record what crosses the env boundary (a wrapper around the reference's bound method; behaviour unchanged)
    log = {"action": [], "reward": [], "done": []}
    real_step = env.step

    def step(actions):
        r, d, info = real_step(actions)
        log["action"].append(np.array(actions, dtype=np.float64).reshape(-1).copy())
        log["reward"].append(float(r)); log["done"].append(bool(d))
        return r, d, info

    env.step = step
    np.random.seed(11)
    th.manual_seed(7)
    stats = []
    for ep in range(3):
        stat = {}
        with contextlib.redirect_stdout(io.StringIO()):
            trainer.behaviour_net.train_process(stat, trainer)
        stats.append({k: float(v) for k, v in stat.items()})
    # model.py:269-306: num_eval_episodes test-mode episodes (tanh(mean), no exploration), the global NumPy stream continuing
    with contextlib.redirect_stdout(io.StringIO()):
        ev = {}
        trainer.behaviour_net.evaluation(ev, trainer)
    g["eval_keys"] = np.array(sorted(ev))
    g["eval"] = np.array([float(ev[k]) for k in sorted(ev)])
    g["eval_steps"] = np.array(len(log["reward"]) - 285)
    # utils/tester.py:16-70: one test-mode episode from a manual reset, recorded through the env's _get_* accessors (env:740-778)
    from utils.tester import PGTester
    with contextlib.redirect_stdout(io.StringIO()):
        record = PGTester(args, trainer.behaviour_net, env).run(3, 7, 1)
    for k, v in record.items():
        g["record." + k] = np.array([np.asarray(x, dtype=np.float64).reshape(-1) for x in v])
    g["steps"] = np.array(trainer.steps)
    g["action"], g["reward"], g["done"] = np.array(log["action"]), np.array(log["reward"]), np.array(log["done"])
    keys = sorted(stats[-1])
    g["stat_keys"] = np.array(keys)
    g["stats"] = np.array([[s.get(k, np.nan) for k in keys] for s in stats])
    for k, v in trainer.behaviour_net.state_dict().items():
        g["final." + k] = v.detach().cpu().numpy().copy()
    import json
    g["alg_args_json"] = np.array(json.dumps(argd, sort_keys=True))
    np.savez_compressed(os.path.join(OUT, "loop_golden.npz"), **g)
    print("wrote loop_golden.npz: steps", trainer.steps, "rewards", np.round(g["reward"][[0, 94, 95, 284]], 5).tolist())
    print("stats", {k: round(stats[-1][k], 5) for k in keys if "loss" in k or "reward" in k})


if __name__ == "__main__":
    main()
