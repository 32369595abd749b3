"""The PCIe-inclusive rate of the env step for a caller that keeps actions and results in HOST memory (as the reference's NumPy
loop does): per vector step, pinned actions -> device, flexenv_step (row-push or stacked observation), reward / done / info (and the
stacked observation) -> pinned host, synchronize.  The C ABI itself takes device pointers; `value` in bench.py is the HBM-resident
rate.  python tools/pcie_probe.py [--steps 200]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--envs", type=int, default=4096)
    a = ap.parse_args()
    import torch
    import safe_marl_amd  # noqa: F401
    from safe_marl_amd.network import create_network
    from safe_marl_amd.series import make_synthetic_series
    from safe_marl_amd.flex_env import VecFlexProvisionEnv
    net = create_network()
    series = make_synthetic_series(net)
    N = a.envs
    for stacked in (False, True):
        env = VecFlexProvisionEnv({}, N, device="cuda:0", net=net, series=series, seed=1234, warm_start=True)
        env.reset()
        h_act = (0.5 + 0.5 * torch.rand(16, N, 5, 4)).float().pin_memory()
        d_act = torch.empty(N, 5, 4, device="cuda")
        h_rew = torch.empty(N, dtype=torch.float64).pin_memory()
        h_don = torch.empty(N, dtype=torch.uint8).pin_memory()
        h_inf = torch.empty_like(env.info, device="cpu").pin_memory()
        h_obs = torch.empty(N, 5, 144).pin_memory() if stacked else None

        def step(k):
            d_act.copy_(h_act[k % 16], non_blocking=True)
            if stacked:
                out = env.step(d_act, fuse_obs=True, auto_reset=True)
            else:
                out = env.step(d_act, obs_rows=True, auto_reset=True)
            r, d, i = out[0], out[1], out[2]
            h_rew.copy_(r, non_blocking=True)
            h_don.copy_(d, non_blocking=True)
            h_inf.copy_(i, non_blocking=True)
            if stacked:
                h_obs.copy_(env.obs, non_blocking=True)
            torch.cuda.synchronize()

        for k in range(20):
            step(k)
        t0 = time.perf_counter()
        for k in range(a.steps):
            step(k)
        dt = (time.perf_counter() - t0) / a.steps
        moved = h_act[0].numel() * 4 + N * 8 + N + h_inf.numel() * 8 + (N * 5 * 144 * 4 if stacked else 0)
        print(f"{'stacked observation to the host' if stacked else 'row push (observation stays on the device)'}: "
              f"{dt * 1e6:7.1f} us per {N}-env step = {N / dt / 1e6:6.1f} M env-steps/s; {moved / 1e6:.2f} MB over PCIe per step "
              f"({moved / dt / 1e9:.1f} GB/s)")
        del env


if __name__ == "__main__":
    main()
