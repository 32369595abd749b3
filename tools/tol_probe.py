"""Env-only step rate against the power-flow tolerance (pf_tol: inf-norm power mismatch in pu).  python tools/tol_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import safe_marl_amd  # noqa: F401
from safe_marl_amd.flex_env import VecFlexProvisionEnv
from safe_marl_amd.network import create_network
from safe_marl_amd.series import make_synthetic_series

net = create_network({})
series = make_synthetic_series(net, n_days=365)
pool = (0.5 + 0.5 * torch.rand(16, 4096, 5, 4, device="cuda")).float()
ref = None
for tol in (1e-12, 1e-10, 1e-9, 1e-8, 1e-7, 1e-6):
    env = VecFlexProvisionEnv({}, 4096, net=net, series=series, seed=1234, warm_start=True, pf_tol=tol)
    env.reset()
    for k in range(48):
        env.step(pool[k % 16], auto_reset=True, obs_rows=True)
    g = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        for k in range(16):
            env.step(pool[k % 16], auto_reset=True, obs_rows=True)
    for _ in range(4):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(64):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (64 * 16)
    v = env.peek("V").double().clone() if hasattr(env, "peek") else None
    sw = env.peek("PF_SWEEPS").float().mean().item()
    # the same trajectory at every tolerance (same seed, same actions, same number of steps): voltages against the tightest
    if ref is None:
        ref = v
        dv = 0.0
    else:
        dv = (v - ref).abs().max().item()
    print(f"pf_tol {tol:.0e}: {us:6.2f} us per launch = {4096 / us:6.1f} M env-steps/s, sweeps {sw:5.2f}, max |V - V(1e-12)| after {48 + 16 * 68} steps {dv:.2e}, Newton steps {env.peek('PF_ITERS').float().mean().item():.3f}")
