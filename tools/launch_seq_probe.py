"""A short timed region (the driver's --steps 20): one HIP graph of 20 step launches (bench.py), twenty Python-level step() calls
and — when the library carries the probe entry `flexenv_step_seq` (a C loop over flexenv_step; built for this measurement in
round 5, found equal to the graph and not kept) — twenty launches enqueued by one C call.  Then the same region on a device
that idled first, with and without a busy spell in front: what bench.py's first timed region sees after its set-up.
    python tools/launch_seq_probe.py"""
import os, sys, time, statistics, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import safe_marl_amd  # noqa: F401
from safe_marl_amd import _lib
from safe_marl_amd.flex_env import VecFlexProvisionEnv, _ptr, _stream
from safe_marl_amd.network import create_network
from safe_marl_amd.series import make_synthetic_series

net = create_network({})
series = make_synthetic_series(net, n_days=365)
env = VecFlexProvisionEnv({}, 4096, net=net, series=series, seed=1234, warm_start=True)
env.reset()
pool = (0.5 + 0.5 * torch.rand(16, 4096, 5, 4, device="cuda")).float()
K = 20
lib = env.lib
vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
HAVE_SEQ = hasattr(lib, "flexenv_step_seq")
if HAVE_SEQ:
    lib.flexenv_step_seq.argtypes = [vp, vp, i32, i64, i32, i32, vp, vp, vp, vp, vp, i32, i32, vp]
    lib.flexenv_step_seq.restype = C.c_int
flags = _lib.FLEX_STEP_AUTORESET | _lib.FLEX_STEP_OBS_ROWS
stride = pool[0].numel() * 4


def seq(n):
    if not HAVE_SEQ:
        for k in range(n):
            env.step(pool[k % 16], auto_reset=True, obs_rows=True)
        return
    _lib.check(lib.flexenv_step_seq(env.handle, _ptr(pool), _lib.FLEX_F32, stride, 16, n, _ptr(env.reward), _ptr(env.done),
                                    _ptr(env.info), _ptr(env.failed), None, 0, flags, _stream()), "flexenv_step_seq")


for k in range(32):
    env.step(pool[k % 16], auto_reset=True, obs_rows=True)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for k in range(K):
        env.step(pool[k % 16], auto_reset=True, obs_rows=True)
torch.cuda.synchronize()


def region(kind):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if kind == "graph":
        g.replay()
    elif kind == "cseq":
        seq(K)
    else:
        for k in range(K):
            env.step(pool[k % 16], auto_reset=True, obs_rows=True)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e6


for kind in (("graph", "cseq", "python") if HAVE_SEQ else ("graph", "python")) * 2:
    for _ in range(20):
        region(kind)
    xs = sorted(region(kind) for _ in range(200))
    print(f"{kind:7s}: median {statistics.median(xs):7.1f} us  p10 {xs[20]:7.1f}  p90 {xs[180]:7.1f}   per step {statistics.median(xs) / K:5.2f} us"
          f"  -> {4096 * K / statistics.median(xs):6.1f} M env-steps/s")

# the same region after the device has idled (what bench.py's first timed region sees after set-up): W = 5 warm-up steps as one
# graph, barrier, K steps — with and without a busy spell in front
g5 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g5):
    for k in range(5):
        env.step(pool[k % 16], auto_reset=True, obs_rows=True)
torch.cuda.synchronize()


def cold_region(idle_ms, busy_steps, events):
    torch.cuda.synchronize()
    time.sleep(idle_ms * 1e-3)
    if busy_steps:
        seq(busy_steps)
    g5.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    if events:
        e0.record()
    g.replay()
    if events:
        e1.record()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) * 1e6
    return dt, (e0.elapsed_time(e1) * 1e3 if events else float("nan"))


for idle_ms, busy, events in ((0, 0, False), (0, 0, True), (50, 0, True), (500, 0, True), (500, 2048, True), (50, 2048, True), (500, 256, True)):
    xs = [cold_region(idle_ms, busy, events) for _ in range(12)]
    wall = sorted(x[0] for x in xs); dev = sorted(x[1] for x in xs)
    print(f"idle {idle_ms:4d} ms, {busy:5d} busy steps, events {int(events)}: wall median {statistics.median(wall):7.1f} us (min {wall[0]:7.1f}), "
          f"event-to-event {statistics.median(dev):7.1f} us -> {4096 * K / statistics.median(wall):6.1f} M env-steps/s")
