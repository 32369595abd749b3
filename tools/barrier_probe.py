"""What the closing barrier of bench.py's bracket costs on RCCL (one rank: all a one-GPU box allows): the driver's 20-step launch
followed by (a) torch.cuda.synchronize() alone, (b) dist.barrier() + synchronize, (c) dist.all_reduce of a cell allocated once +
synchronize (what ProcessGroupNCCL::barrier does, minus the fill launch for a fresh tensor).  Interleaved repetitions, wall clock.
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port P tools/barrier_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import torch.distributed as dist
    import safe_marl_amd  # noqa: F401
    from safe_marl_amd.network import create_network
    from safe_marl_amd.series import make_synthetic_series
    from safe_marl_amd.flex_env import VecFlexProvisionEnv
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", device_id=dev)
    net = create_network()
    series = make_synthetic_series(net)
    env = VecFlexProvisionEnv({}, 4096, device="cuda:0", net=net, series=series, seed=1234, warm_start=True)
    env.reset()
    gen = torch.Generator(device="cuda").manual_seed(99)
    pool = (0.5 + 0.5 * torch.rand(16, 4096, 5, 4, device="cuda", generator=gen)).float()
    launch = env.step_many_prepared(pool, steps=20, auto_reset=True)[0]
    big = env.step_many_prepared(pool, steps=256, auto_reset=True)[0]
    cell = torch.zeros(1, device=dev)
    dist.barrier(); dist.all_reduce(cell); torch.cuda.synchronize()
    for _ in range(8):
        big()
    torch.cuda.synchronize()
    res = {"sync": [], "barrier": [], "cell": []}
    for r in range(150):
        for mode in ("sync", "barrier", "cell"):
            big()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            launch()
            if mode == "barrier":
                dist.barrier()
            elif mode == "cell":
                dist.all_reduce(cell)
            torch.cuda.synchronize()
            res[mode].append((time.perf_counter() - t0) * 1e6)
    for k, v in res.items():
        v.sort()
        print(f"{k:8s}: median {v[len(v) // 2]:8.2f} us   p10 {v[len(v) // 10]:8.2f}   p90 {v[9 * len(v) // 10]:8.2f}")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
