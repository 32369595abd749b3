"""CPU baseline scaling of the C restatement with OpenMP threads, plus the cgroup CPU quota of the box."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time, numpy as np, subprocess
if len(sys.argv) > 1:
    th = int(sys.argv[1]); os.environ["OMP_NUM_THREADS"] = str(th)
    import safe_marl_amd
    from safe_marl_amd.network import create_network
    from safe_marl_amd.series import make_synthetic_series
    from oracle import c_oracle
    net = create_network(); s = make_synthetic_series(net, n_days=60); rng = np.random.default_rng(0)
    n = 4096
    e = c_oracle.COracleEnv(net, s.table, n)
    day = rng.integers(0, s.n_start_days(96), n); start = rng.integers(0, 4, n) + rng.integers(0, 24, n) * 4 + day * 96
    e.reset(start, rng.uniform(0.01125, 0.01375, (n, 5)), rng.uniform(0, 1, (n, 20)))
    acts = rng.uniform(0.5, 1, (4, n, 5, 4)); e.step(acts[0])
    t = time.perf_counter(); K = 10
    for k in range(K): e.step(acts[k % 4])
    print(f"threads {th:4d}: {n*K/(time.perf_counter()-t):10.0f} env-steps/s")
else:
    for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
        try: print(p, open(p).read().strip())
        except Exception as ex: print(p, "n/a")
    print("affinity", len(os.sched_getaffinity(0)))
    for th in (1, 8, 16, 32, 64, 128, 256):
        print(subprocess.run([sys.executable, __file__, str(th)], capture_output=True, text=True).stdout.strip())
