"""RCCL itself on the one-GPU box: a world-size-1 NCCL process group (watchdog thread and all), the trainer told it has two
ranks so that its sub-updates take the multi-rank route — graph A, ncclAllReduce of the flat bucket (an identity at world
size 1), graph B — for one training episode.  Checks what the gloo rehearsal cannot: HIP-graph capture next to a live NCCL
communicator and its watchdog, and RCCL collectives issued between graph replays."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "examples"))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import numpy as np, torch, torch.distributed as dist
import safe_marl_amd
from train_maddpg import DEFAULT_ALG_ARGS
from safe_marl_amd.flex_env import VecFlexProvisionEnv
from safe_marl_amd.learner import MADDPG
from safe_marl_amd.network import create_network
from safe_marl_amd.series import make_synthetic_series
from safe_marl_amd.trainer import PGTrainer
from safe_marl_amd.util import convert
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
net = create_network(); series = make_synthetic_series(net, n_days=30)
alg = dict(DEFAULT_ALG_ARGS); alg.update(alg="maddpg", agent_num=5, obs_size=144, state_size=110, action_dim=4, behaviour_update_freq=30, target_update_freq=60)
env = VecFlexProvisionEnv({}, 1024, net=net, series=series, seed=1, warm_start=True)
torch.manual_seed(0); np.random.seed(0)
tr = PGTrainer(convert(alg), MADDPG, env, None, replay_capacity=1024 * 96 * 2)
tr.world = 2                       # take the multi-rank route: split graphs around an all-reduce (sum over ONE rank, then x 1/2)
w0 = torch.cat([p.detach().reshape(-1) for p in tr.behaviour_net.parameters()]).clone()
for ep in range(3):
    stat = {}
    tr.behaviour_net.train_process(stat, tr)
torch.cuda.synchronize()
w1 = torch.cat([p.detach().reshape(-1) for p in tr.behaviour_net.parameters()])
assert torch.isfinite(w1).all() and not torch.equal(w0, w1)
assert set(tr._update_graphs) == {"value", "policy"}
fused = [bool(g["allreduce_in_graph"]) for g in tr._update_graphs.values()]
assert all(fused) or all(g["apply"] is not None for g in tr._update_graphs.values())
print("nccl world-1 probe ok:", "ONE graph per sub-update with ncclAllReduce captured inside it" if all(fused) else
      "split update graphs with ncclAllReduce between them", "-", tr.steps, "vector steps, value loss",
      float(stat["mean_train_value_loss"]), "ALLREDUCE_IN_GRAPH" if all(fused) else "SPLIT_GRAPHS")
dist.barrier(); dist.destroy_process_group()
