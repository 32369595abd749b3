"""Safety layer: time of flexenv_safety_project at 8192 envs and of the whole safety-signal pipeline."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time, numpy as np, torch
import safe_marl_amd
from safe_marl_amd.network import create_network
from safe_marl_amd.series import make_synthetic_series
from safe_marl_amd.flex_env import VecFlexProvisionEnv
from safe_marl_amd import safety_signal as ss
net = create_network(); s = make_synthetic_series(net, n_days=60)
N = 8192
env = VecFlexProvisionEnv({"alg": "safemaddpg"}, N, net=net, series=s)
env.reset()
t = time.perf_counter(); vp = ss.fit_voltage_predictor(net, num_scenarios=1000); torch.cuda.synchronize()
print(f"safety-signal pipeline (1000 scenarios: batched HIP power flow + OLS fit): {(time.perf_counter()-t)*1e3:.1f} ms, test MSE {vp.test_mse:.2e}")
sp, sq, beta = vp.building_terms(net)
prop = torch.rand(N, 5, 4, device="cuda")
for _ in range(10): env.safety_project(prop, sp, sq, beta, 0.9, 1.1)
torch.cuda.synchronize(); t = time.perf_counter(); K = 200
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
spd, sqd, bd = (torch.as_tensor(x, dtype=torch.float64, device="cuda") for x in (sp, sq, beta))
e0.record()
for _ in range(K): adj, hit = env.safety_project(prop, spd, sqd, bd, 0.9, 1.1)
e1.record(); torch.cuda.synchronize()
print(f"flexenv_safety_project: {e0.elapsed_time(e1)/K*1e3:.1f} us per call for {N} envs x 5 buildings ({N*5/(e0.elapsed_time(e1)/K*1e-3)/1e9:.2f} G projections/s), intervened {hit.float().mean().item():.2f}")
