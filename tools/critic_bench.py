"""Fused critic tail (csrc/critic.hip) vs the PyTorch module + autograd it replaces: forward and forward+backward on
the update's row count.  HIP-event timed."""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import safe_marl_amd  # noqa: F401
from safe_marl_amd.nets import MLPCritic

args = types.SimpleNamespace(hid_size=64, layernorm=True, hid_activation="relu")
c = MLPCritic(745, 1, args).cuda()
params = [p for n, p in c.named_parameters() if not n.startswith("fc1")]


def timed(fn, n=100):
    for _ in range(10):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for rows in (20480, 163840):
    z = torch.randn(rows, 64, device="cuda", requires_grad=True)
    w = torch.randn(rows, 1, device="cuda")

    def fwd():
        with torch.no_grad():
            c.forward_from_hidden(z, need_hidden=False)

    def fwdbwd():
        q, _ = c.forward_from_hidden(z, need_hidden=False)
        torch.autograd.grad((q * w).sum(), [z] + params)

    out = []
    for fused in (True, False):
        c.fused_tail = fused
        out.append((timed(fwd), timed(fwdbwd)))
    print(f"{rows:7d} rows: forward fused {out[0][0]:6.1f} us / module {out[1][0]:6.1f} us;  forward+backward fused "
          f"{out[0][1]:6.1f} us / autograd {out[1][1]:6.1f} us")

# matrix-core (variant 0) vs VALU (variant 1) kernels: forward and the dz1-only backward, HIP-graph replays
from safe_marl_amd import nets
for p in c.parameters():
    p.requires_grad_(False)
c.fused_tail = True


def graph_timed(fn, reps=20):
    for _ in range(3):
        fn()
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                fn()
    import time
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / (10 * reps) * 1e6


for rows in (20480, 163840):
    z = torch.randn(rows, 64, device="cuda", requires_grad=True)
    w = torch.randn(rows, 1, device="cuda")
    for variant in (0, 1):
        nets.CRITIC_VARIANT = variant

        def fwd():
            with torch.no_grad():
                nets.CriticTail.apply(z, c)

        def both():
            torch.autograd.grad((nets.CriticTail.apply(z, c) * w).sum(), [z])

        tf, tb = graph_timed(fwd), graph_timed(both)
        print(f"{rows:7d} rows, variant {variant}: forward {tf:6.1f} us, forward + dz1-only backward {tb:6.1f} us")
    nets.CRITIC_VARIANT = 0

# the value sub-update's critic backward (flexnet_critic_td_backward: statistics, matrix-core backward with parameter
# gradients forming the TD error itself, dz fold, finish) on 16-row tiles / two wavefronts per SIMD vs the 32-row kernel
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
for v32 in (0, 2, 1):
    nets.CRITIC_PGRAD32 = v32
    r = bench.critic_backward_roofline(lambda fn, n=200: timed(fn, n) * 1e-6)
    print(f"td backward, {['16-row tiles, sample-major, 2 wavefronts/SIMD', '32-row tiles + fold, 1 wavefront/SIMD', '16-row tiles + fold, 2 wavefronts/SIMD'][v32]}: "
          f"{r['launch_us']:6.1f} us for {r['rows']} rows = {r['achieved']:.1f} TFLOP/s fp32 = {r['frac']:.3f} of the matrix peak")
nets.CRITIC_PGRAD32 = 0

