"""Kernel-level profile of one value sub-update and one policy sub-update at the vectorised batch (32 x N/4 rows)."""
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
import torch.profiler as P
net = create_network(); series = make_synthetic_series(net, n_days=100)
N = 4096
env = VecFlexProvisionEnv({}, N, net=net, series=series, warm_start=True)
alg = dict(DEFAULT_ALG_ARGS); alg.update(alg="maddpg", agent_num=5, obs_size=144, state_size=110, action_dim=4)
a = dict(alg); a["behaviour_update_freq"] = 10**9; a["target_update_freq"] = 10**9
tr = PGTrainer(convert(a), MADDPG, env, None, replay_capacity=N * 96 * 2)
st = {}
tr.behaviour_net.train_process(st, tr); torch.cuda.synchronize()
print("replay length", tr.replay_buffer.length, "effective batch", tr.effective_batch_size())
for which in ("value", "policy"):
    fn = tr.value_replay_process if which == "value" else tr.policy_replay_process
    for _ in range(3):
        fn(st)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10):
        fn(st)
    torch.cuda.synchronize(); print(which, "sub-update", (time.perf_counter() - t) / 10 * 1e3, "ms")
    with P.profile(activities=[P.ProfilerActivity.CPU, P.ProfilerActivity.CUDA]) as prof:
        for _ in range(5):
            fn(st)
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=22, max_name_column_width=70))
