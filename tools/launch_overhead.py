"""Is bench.py's loop bound by the host (Python + ctypes launch) or by the kernel?  Issue time vs completion time."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time, torch
import safe_marl_amd
from safe_marl_amd.network import create_network
from safe_marl_amd.series import make_synthetic_series
from safe_marl_amd.flex_env import VecFlexProvisionEnv
net = create_network(); s = make_synthetic_series(net, n_days=100)
for N in (256, 4096):
    env = VecFlexProvisionEnv({}, N, net=net, series=s, warm_start=True)
    pool = (0.5 + 0.5 * torch.rand(16, N, 5, 4, device="cuda")).float()
    env.reset()
    for k in range(200): env.step(pool[k % 16], fuse_obs=True, auto_reset=True)
    torch.cuda.synchronize()
    K = 2000
    t0 = time.perf_counter()
    for k in range(K): env.step(pool[k % 16], fuse_obs=True, auto_reset=True)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"N={N:5d}: issue {1e6*(t1-t0)/K:6.2f} us/step, complete {1e6*(t2-t0)/K:6.2f} us/step")
