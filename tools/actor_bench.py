"""Fused actor inference (csrc/actor.hip) vs the PyTorch module it replaces, on the rollout's and the update's row
counts.  HIP-event timed, 200 launches each."""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import safe_marl_amd  # noqa: F401
from safe_marl_amd.nets import RNNAgent, fused_actor_forward

args = types.SimpleNamespace(hid_size=64, layernorm=True, action_dim=4, agent_num=5, hid_activation="relu")
agent = RNNAgent(149, args).cuda()
ids = torch.eye(5, device="cuda")


def timed(fn, n=200):
    for _ in range(20):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for b in (4096, 32768):
    obs = torch.randn(b, 5, 144, device="cuda"); hid = torch.randn(b, 5, 64, device="cuda")
    with torch.no_grad():
        t_f = timed(lambda: fused_actor_forward(agent, obs, hid, 5, True))
        t_v = timed(lambda: fused_actor_forward(agent, obs, hid, 5, True, variant=1))
        t_m = timed(lambda: agent(torch.cat((obs, ids.expand(b, -1, -1)), -1).reshape(b * 5, -1), hid.reshape(b * 5, -1)))
    macs = b * 5 * (144 * 64 + 2 * 192 * 64 + 64 * 4)
    print(f"{b * 5:7d} rows: MFMA kernel {t_f:7.1f} us ({2 * macs / t_f / 1e6:5.1f} TFLOP/s fp32)   VALU kernel {t_v:7.1f} us   "
          f"module {t_m:7.1f} us   x{t_m / t_f:.1f}")

# rollout-size batches: the five-tiles-per-CU kernel on 16-row tiles (variant 2) against the 32-row kernel (variant 3).
# HIP-graph replays of 20 calls: an eager call costs the host ~20 us of Python + ctypes, more than the kernel takes
def graph_timed(fn, reps=20):
    import time
    for _ in range(3):
        fn()
    g = torch.cuda.CUDAGraph(); st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        fn()
        with torch.cuda.graph(g, stream=st):
            for _ in range(reps):
                fn()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / (20 * reps) * 1e6


print("rows: five 16-row tiles per CU (wavefronts 0-3 one tile each, 4-7 share the fifth) | 32-row tiles, two wavefronts per SIMD")
for b, n in ((1024, 5), (2048, 5), (4096, 3), (4096, 5), (6144, 5), (8192, 5), (16384, 5), (32768, 5)):
    obs = torch.randn(b, n, 144, device="cuda"); hid = torch.randn(b, n, 64, device="cuda")
    ag = agent if n == 5 else RNNAgent(144 + n, types.SimpleNamespace(hid_size=64, layernorm=True, action_dim=4, agent_num=n,
                                                                      hid_activation="relu")).cuda()
    with torch.no_grad():
        t16 = graph_timed(lambda: fused_actor_forward(ag, obs, hid, n, True, variant=2))
        t32 = graph_timed(lambda: fused_actor_forward(ag, obs, hid, n, True, variant=3))
    print(f"{b * n:7d} rows: {t16:7.1f} us | {t32:7.1f} us", flush=True)

# what the rollout adds to the 20 480-row call: the exploration epilogue with in-kernel noise, and cold caches (the env step
# between two policy calls streams ~40 MB: emulated by a 64 MB copy between calls)
b, n = 4096, 5
obs = torch.randn(b, n, 144, device="cuda"); hid = torch.randn(b, n, 64, device="cuda")
rng = torch.tensor([1234, 0], dtype=torch.int64, device="cuda")
src, dst = torch.randn(16 << 20, device="cuda"), torch.empty(16 << 20, device="cuda")
with torch.no_grad():
    t_plain = graph_timed(lambda: fused_actor_forward(agent, obs, hid, n, True, variant=2))
    t_noise = graph_timed(lambda: fused_actor_forward(agent, obs, hid, n, True, variant=2, rng_state=rng))
    nz = torch.randn(b, n, 4, device="cuda")
    t_tensor_noise = graph_timed(lambda: fused_actor_forward(agent, obs, hid, n, True, variant=2, noise=nz))
    t_copy = graph_timed(lambda: dst.copy_(src))
    t_cold = graph_timed(lambda: (dst.copy_(src), fused_actor_forward(agent, obs, hid, n, True, variant=2, rng_state=rng)))
print(f"20480 rows, five-tiles-per-CU kernel: plain {t_plain:.1f} us; exploration epilogue on a noise tensor {t_tensor_noise:.1f} us; "
      f"with in-kernel exploration noise {t_noise:.1f} us; "
      f"behind a 64 MB copy ({t_copy:.1f} us) {t_cold - t_copy:.1f} us")
