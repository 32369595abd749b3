"""Rollout-only and rollout+update time per vector step, eager vs HIP-graph rollout."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time, torch
import safe_marl_amd
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
from train_maddpg import DEFAULT_ALG_ARGS
from safe_marl_amd.flex_env import VecFlexProvisionEnv
from safe_marl_amd.learner import MADDPG
from safe_marl_amd.network import create_network
from safe_marl_amd.series import make_synthetic_series
from safe_marl_amd.trainer import PGTrainer
from safe_marl_amd.util import convert
net = create_network(); series = make_synthetic_series(net, n_days=100)
N = 4096
env = VecFlexProvisionEnv({}, N, net=net, series=series, warm_start=True)
alg = dict(DEFAULT_ALG_ARGS); alg.update(alg="maddpg", agent_num=5, obs_size=144, state_size=110, action_dim=4)
for graph in (False, True):
    for freq, label in ((10**9, "rollout only     "), (60, "rollout + updates")):
        a = dict(alg); a["behaviour_update_freq"] = freq; a["target_update_freq"] = 2 * freq
        tr = PGTrainer(convert(a), MADDPG, env, None, replay_capacity=N * 96 * 2, graph_rollout=graph)
        st = {}
        tr.behaviour_net.train_process(st, tr); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(3): tr.behaviour_net.train_process(st, tr)
        torch.cuda.synchronize()
        print(f"graph={graph!s:5s} {label} {(time.perf_counter() - t) / 285 * 1e3:.3f} ms per vector step")
