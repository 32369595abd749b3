"""Long training soak through trainer.run (rollout graphs, graphed sub-updates, target updates, evaluation every 20 episodes):
prints a line every 50 episodes; fails on a non-finite statistic."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "examples"))
import numpy as np, torch
import safe_marl_amd
from train_maddpg import DEFAULT_ALG_ARGS
from safe_marl_amd import learner
from safe_marl_amd.flex_env import VecFlexProvisionEnv
from safe_marl_amd.network import create_network
from safe_marl_amd.series import make_synthetic_series
from safe_marl_amd.trainer import PGTrainer
from safe_marl_amd.util import convert
alg_name = sys.argv[1] if len(sys.argv) > 1 else "maddpg"
episodes = int(sys.argv[2]) if len(sys.argv) > 2 else 300
envs = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
env_args = {"alg": "safemaddpg"} if alg_name == "safemaddpg" else {}
net = create_network(env_args); series = make_synthetic_series(net, n_days=365)
env = VecFlexProvisionEnv(env_args, envs, net=net, series=series, seed=1, warm_start=True)
a = dict(DEFAULT_ALG_ARGS); a.update(alg=alg_name, agent_num=5, obs_size=144, state_size=110, action_dim=4, v_min=0.9, v_max=1.1)
cls = {"maddpg": learner.MADDPG, "safemaddpg": learner.SAFEMADDPG, "matd3": learner.MATD3, "iddpg": learner.IDDPG}[alg_name]
if os.environ.get("SOAK_UNFUSED_TD") == "1":          # A/B: the forward -> flexnet_td_loss -> backward sequence
    learner.MADDPG.fused_td_backward = False
if os.environ.get("SOAK_NO_BURSTS") == "1":
    learner.RolloutGraph.BURSTS = ()
torch.manual_seed(0); np.random.seed(0)
tr = PGTrainer(convert(a), cls, env, None, replay_capacity=envs * 96 * 2)
t0 = time.perf_counter()
for ep in range(episodes):
    stat = {}
    tr.run(stat, ep)
    bad = [k for k, v in stat.items() if not np.isfinite(v)]
    assert not bad, (ep, bad, stat)
    if ep % 50 == 0 or ep == episodes - 1:
        print(f"{alg_name} ep {ep:4d} reward {stat['mean_train_reward']:+.5f} vloss {stat.get('mean_train_value_loss', float('nan')):.4f} "
              f"ploss {stat.get('mean_train_policy_loss', float('nan')):+.4f} vpen {stat['mean_train_voltage_penalty']:.5f} "
              f"gaps {len(tr.replay_buffer.gaps)} k {tr.replay_buffer.k} t {time.perf_counter() - t0:.1f}s", flush=True)
w = torch.cat([p.detach().reshape(-1) for p in tr.behaviour_net.parameters()])
assert torch.isfinite(w).all()
print("soak ok", alg_name, episodes, "episodes", tr.steps, "vector steps")
