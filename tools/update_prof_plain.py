"""20 value and 20 policy sub-updates (HIP-graph replays incl. the replay-window refresh) after one rollout episode,
for `rocprofv3 --kernel-trace --stats` (no torch.profiler inside)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "examples"))
import safe_marl_amd  # noqa: E402,F401
from train_maddpg import DEFAULT_ALG_ARGS  # noqa: E402
from safe_marl_amd.flex_env import VecFlexProvisionEnv  # noqa: E402
from safe_marl_amd import learner  # noqa: E402
from safe_marl_amd.network import create_network  # noqa: E402
from safe_marl_amd.series import make_synthetic_series  # noqa: E402
from safe_marl_amd.trainer import PGTrainer  # noqa: E402
from safe_marl_amd.util import convert  # noqa: E402

net = create_network()
series = make_synthetic_series(net, n_days=100)
N = int(os.environ.get("ENVS", "4096"))
env = VecFlexProvisionEnv({}, N, net=net, series=series, warm_start=True)
alg = dict(DEFAULT_ALG_ARGS)
ALG = os.environ.get("ALG", "maddpg")
MADDPG = {"maddpg": learner.MADDPG, "matd3": learner.MATD3, "iddpg": learner.IDDPG}[ALG]
alg.update(alg=ALG, agent_num=5, obs_size=144, state_size=110, action_dim=4, behaviour_update_freq=10 ** 9,
           target_update_freq=10 ** 9)
tr = PGTrainer(convert(alg), MADDPG, env, None, replay_capacity=N * 96 * 2)
st = {}
tr.behaviour_net.train_process(st, tr)
torch.cuda.synchronize()
for which in ("value", "policy"):
    fn = tr.value_replay_process if which == "value" else tr.policy_replay_process
    for _ in range(3):
        fn(st)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(20):
        fn(st)
    torch.cuda.synchronize()
    print(which, "sub-update", (time.perf_counter() - t) / 20 * 1e3, "ms")

# EVENTS=n [BATCH_DIV=d]: n whole update events (ten value sub-updates + one policy sub-update, model.py:47-50) at
# batch_scale = N / d through trainer.replay_event — with d = 1 (the reference's sample reuse) the bootstrap values are filed
# once per event for the union of the windows (FLEX_BOOTSTRAP_CACHE=0: per sub-update)
if os.environ.get("EVENTS"):
    import numpy as np
    tr.batch_scale = max(1, N // int(os.environ.get("BATCH_DIV", "1")))
    for _ in range(int(os.environ.get("FILL_EPISODES", "1"))):
        tr.behaviour_net.train_process(st, tr)
    np.random.seed(0)
    for _ in range(2):
        tr.replay_event(st, 10, 1)
    torch.cuda.synchronize()
    n_ev = int(os.environ["EVENTS"])
    t = time.perf_counter()
    for _ in range(n_ev):
        tr.replay_event(st, 10, 1)
    torch.cuda.synchronize()
    print(f"update event at batch {tr.effective_batch_size()}: {(time.perf_counter() - t) / n_ev * 1e3:.3f} ms "
          f"({tr.bootstrap_cached_events} of {n_ev + 2} events on filed bootstrap values)")
