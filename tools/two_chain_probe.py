"""How much would two independent half-batch rollout chains on two streams buy over one full-batch chain?  (Both kernels
of a rollout step are latency-bound at 4096 envs: the actor runs one 32-row tile per busy wavefront, the env kernel two
wavefronts per SIMD.)  Rollout only, no updates; two separate env objects / rings of 2048 envs against one of 4096."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "examples"))
import torch
import safe_marl_amd
from train_maddpg import DEFAULT_ALG_ARGS
from safe_marl_amd.flex_env import VecFlexProvisionEnv
from safe_marl_amd.learner import MADDPG, RolloutGraph
from safe_marl_amd.network import create_network
from safe_marl_amd.replay_buffer import TransReplayBuffer
from safe_marl_amd.series import make_synthetic_series
from safe_marl_amd.util import convert
net = create_network(); series = make_synthetic_series(net, n_days=60)
alg = dict(DEFAULT_ALG_ARGS); alg.update(alg="maddpg", agent_num=5, obs_size=144, state_size=110, action_dim=4)
torch.manual_seed(0)
m = MADDPG(convert(alg)).cuda()

def chain(n, seed):
    env = VecFlexProvisionEnv({}, n, net=net, series=series, seed=seed, warm_start=True)
    rg = RolloutGraph(m, env, TransReplayBuffer(n * 64, device="cuda"))
    rg.start_episode(env.reset()); rg.capture(); rg.start_episode(env.reset())
    return rg

def timed(fn, steps=950):
    for _ in range(95): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(steps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / steps * 1e6

one = chain(4096, 1)
print("one chain of 4096 envs: %.1f us per vector step" % timed(lambda: one.step()))
a, b = chain(2048, 2), chain(2048, 3)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
def both():
    with torch.cuda.stream(sa): a.step()
    with torch.cuda.stream(sb): b.step()
print("two chains of 2048 envs on two streams: %.1f us per vector step (4096 env-steps)" % timed(both))
def both_serial():
    a.step(); b.step()
print("two chains of 2048 envs on one stream: %.1f us per vector step" % timed(both_serial))
q = [chain(1024, 10 + i) for i in range(4)]; sq = [torch.cuda.Stream() for _ in range(4)]
def four():
    for r, s in zip(q, sq):
        with torch.cuda.stream(s): r.step()
print("four chains of 1024 envs on four streams: %.1f us per vector step" % timed(four))
