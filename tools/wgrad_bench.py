"""Time csrc/wgrad.hip against the library's dY^T X on the update's shapes (HIP-graph replays, so launch overhead is out)."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time, torch
import safe_marl_amd
from safe_marl_amd.nets import tall_wgrad


def bench(f, reps=20):
    for _ in range(3): f()
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        f()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10): g.replay()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / (10 * reps) * 1e6


def bench_eager(f, reps=10):
    for _ in range(2): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for k, m, n in [(163840, 64, 149), (163840, 192, 64), (163840, 4, 64), (32768, 64, 720), (32768, 64, 20), (20480, 64, 149)]:
    dy = torch.randn(k, m, device="cuda"); x = torch.randn(k, n, device="cuda")
    print("k=%d m=%d n=%d" % (k, m, n), end=" ", flush=True)
    t_hip = bench(lambda: tall_wgrad(dy, x)); print("hip done", end=" ", flush=True)
    t_lib = bench_eager(lambda: dy.t() @ x); print("library done", flush=True)
    gb = k * (m + n) * 4 / 1e9; fl = 2.0 * k * m * n / 1e12
    print("k=%6d m=%3d n=%3d  hip %7.1f us (%5.2f TB/s, %5.1f TFLOP/s)   library %7.1f us" % (k, m, n, t_hip, gb / t_hip * 1e3, fl / t_hip * 1e6, t_lib))

# the critic's first layer as the value sub-update calls it: [obs block | action block] as ONE operand (two views), column sums
k, m = 32768, 64
dy = torch.randn(k, m, device="cuda"); x = torch.randn(k, 720, device="cuda"); x2 = torch.randn(k, 20, device="cuda")
out = torch.empty(m, 745, device="cuda"); cs = torch.empty(m, device="cuda")
t_two = bench(lambda: tall_wgrad(dy, x, out=out[:, :720], colsum=cs, x2=x2, out2=out[:, 725:745]))
print("k=%6d m=%3d n=720+20 (two views, column sums)  hip %7.1f us" % (k, m, t_two))
