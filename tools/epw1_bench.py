"""(Round-3 experiment, kept for the record: the library now ships the 256-register build only, so the three modes below
measure the same kernel; the numbers that decided it are in profiles/r03_epw1_bench.txt.)

One environment per wavefront (feeders with more than 32 PQ buses): the step kernel built for four wavefronts per
SIMD (128 registers, ~310 B of scratch per lane) against the build for two (256 registers, no scratch), at batch sizes
on both sides of the two-per-SIMD residency limit (8 environments per CU).  HIP-event timed, HIP-graph replays of 16 steps."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import safe_marl_amd  # noqa: F401
from safe_marl_amd import _lib
from safe_marl_amd.flex_env import VecFlexProvisionEnv
from safe_marl_amd.series import make_synthetic_series
from tests.test_pf_gpu import _random_feeder

lib = _lib.load()
has_switch = hasattr(lib, "flexenv_debug_set_epw1_wide")
if has_switch:
    lib.flexenv_debug_set_epw1_wide.argtypes = [C.c_int]
blds = [7, 19, 33, 41]
net = _random_feeder(45, 11, blds)
series = make_synthetic_series(net, n_days=60)
for n in (1024, 2048, 3072, 4096, 8192):
    env = VecFlexProvisionEnv({"buildings": blds, "pv_nodes": blds, "ess_nodes": blds}, n, series=series, net=net, seed=3,
                              warm_start=True)
    pool = (0.5 + 0.5 * torch.rand(16, n, 4, 4, device="cuda")).float()
    res = {}
    for mode, name in ((0, "128-register build (4 wavefronts/SIMD, spills)"), (1, "256-register build (2 wavefronts/SIMD)"), (-1, "auto")):
        if has_switch:
            lib.flexenv_debug_set_epw1_wide(mode)
        env.reset()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            env.step(pool[0], fuse_obs=True, auto_reset=True)
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for k in range(16):
                env.step(pool[k], fuse_obs=True, auto_reset=True)
        for _ in range(4):
            g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(32):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) / 512 * 1e3
        assert float(env.failed.float().mean().item()) < 0.01
    if has_switch:
        lib.flexenv_debug_set_epw1_wide(-1)
    print(f"45-bus feeder, {n:5d} envs: " + "; ".join(f"{k}: {v:6.1f} us/step" for k, v in res.items()), flush=True)
    del env
