"""Which ATen kernels do the eager sub-updates of MATD3 / IDDPG launch (candidates for graph capture)?"""
import os, sys
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
from safe_marl_amd.util import convert, GRAPH_DENYLIST
from torch.profiler import ProfilerActivity, profile
net = create_network(); series = make_synthetic_series(net, n_days=30)
for alg in ("matd3", "iddpg"):
    a = dict(DEFAULT_ALG_ARGS); a.update(alg=alg, agent_num=5, obs_size=144, state_size=110, action_dim=4, behaviour_update_freq=10**9, target_update_freq=10**9)
    env = VecFlexProvisionEnv({}, 4096, net=net, series=series, seed=3, warm_start=True)
    tr = PGTrainer(convert(a), {"matd3": learner.MATD3, "iddpg": learner.IDDPG}[alg], env, None, replay_capacity=4096 * 96 * 2)
    tr.behaviour_net.train_process({}, tr)
    for which in ("value", "policy"):
        fn = tr.value_replay_process if which == "value" else tr.policy_replay_process
        fn({}); torch.cuda.synchronize()
        import time
        t = time.perf_counter()
        for _ in range(5): fn({})
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5
        with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
            fn({}); torch.cuda.synchronize()
        ev = [(e.key, e.count, e.self_device_time_total) for e in prof.key_averages() if "cuda" in str(getattr(e, "device_type", "")).lower()]
        tot = sum(x[2] for x in ev)
        print(f"== {alg} {which}: {dt*1e3:.3f} ms wall, {tot/1e3:.3f} ms GPU, {sum(x[1] for x in ev)} kernels")
        for k, c, t_ in sorted(ev, key=lambda x: -x[2])[:14]:
            flag = "DENY" if any(d in k for d in GRAPH_DENYLIST) else "    "
            print(f"   {flag} {t_:8.1f} us x{c:3d} {k[:110]}")
        for k, c, t_ in ev:
            if any(d in k for d in GRAPH_DENYLIST): print("   DENYLISTED:", c, t_, k[:140])
