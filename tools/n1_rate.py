"""PCIe-inclusive rate of the N=1 drop-in view (host NumPy in/out every step)."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time, numpy as np, torch
import safe_marl_amd
from safe_marl_amd.network import create_network
from safe_marl_amd.series import make_synthetic_series
from safe_marl_amd.flex_env import FlexibilityProvisionEnv
net = create_network(); s = make_synthetic_series(net, n_days=30)
env = FlexibilityProvisionEnv({"seed": 0}, net=net, series=s)
rng = np.random.default_rng(0)
env.reset()
t = time.perf_counter(); n = 0
for ep in range(5):
    env.reset()
    for k in range(95):
        env.step(rng.uniform(0.5, 1, (5, 4)).astype(np.float32)); env.get_obs(); n += 1
dt = time.perf_counter() - t
print(f"N=1 drop-in view: {n/dt:.0f} env-steps/s ({dt/n*1e6:.0f} us per step()+get_obs(), host numpy in/out)")
