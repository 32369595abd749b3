"""Where the vectorised training loop spends its time: rollout only vs rollout + updates, and a torch profile."""
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
def run(freq, label):
    a = dict(alg); a["behaviour_update_freq"] = freq; a["target_update_freq"] = 2 * freq
    tr = PGTrainer(convert(a), MADDPG, env, None, replay_capacity=N * 96 * 2)
    st = {}
    tr.behaviour_net.train_process(st, tr); torch.cuda.synchronize()
    t = time.perf_counter(); tr.behaviour_net.train_process(st, tr); tr.behaviour_net.train_process(st, tr); torch.cuda.synchronize()
    print(label, (time.perf_counter() - t) / 190 * 1e3, "ms per vector step")
    return tr
tr = run(10**9, "rollout only      ")
run(60, "rollout + updates ")
# micro: one update at the effective batch
import torch.profiler as P
tr = run(60, "again             ")
st = {}
with P.profile(activities=[P.ProfilerActivity.CPU, P.ProfilerActivity.CUDA]) as prof:
    tr.behaviour_net.args = tr.args
    for _ in range(20):
        pass
    tr.steps = 1
    m = tr.behaviour_net
    obs = env.reset().clone(); hid = torch.zeros(N, 5, 64, device="cuda"); avail = torch.ones(N, 5, 4, device="cuda")
    for t in range(10):
        with torch.no_grad():
            action, action_pol, lp, _, hid2 = m.get_actions(obs, status="train", exploration=True, actions_avail=avail, target=False, last_hid=hid)
            actual = m.env_action(action)
        env.step(actual, fuse_obs=True, auto_reset=True)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=14, max_name_column_width=60))
