"""Host-side cost of a short timed region (the driver's --steps 20): one graph of 20 env-step launches, timed as bench.py does
(HIP events inside the bracket) / without the events / with a stream-query spin in front of the synchronize.
    python tools/launch_region_probe.py"""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import safe_marl_amd  # noqa: F401
from safe_marl_amd.flex_env import VecFlexProvisionEnv
from safe_marl_amd.network import create_network
from safe_marl_amd.series import make_synthetic_series

net = create_network({})
series = make_synthetic_series(net, n_days=365)
env = VecFlexProvisionEnv({}, 4096, net=net, series=series, seed=1234, warm_start=True)
env.reset()
pool = (0.5 + 0.5 * torch.rand(16, 4096, 5, 4, device="cuda")).float()
K = 20
for k in range(32):
    env.step(pool[k % 16], auto_reset=True, obs_rows=True)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for k in range(K):
        env.step(pool[k % 16], auto_reset=True, obs_rows=True)
torch.cuda.synchronize()
st = torch.cuda.current_stream()


def region(kind):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if kind == "events":
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record()
    else:
        g.replay()
    if kind == "spin":
        while not st.query():
            pass
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e6


for kind in ("events", "plain", "spin", "events", "plain", "spin"):
    for _ in range(20):
        region(kind)
    xs = sorted(region(kind) for _ in range(200))
    print(f"{kind:7s}: median {statistics.median(xs):7.1f} us  p10 {xs[20]:7.1f}  p90 {xs[180]:7.1f}   per step {statistics.median(xs) / K:5.2f} us")
