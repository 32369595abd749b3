"""Phase breakdown of one wavefront of the critic's matrix-core backward with parameter gradients (csrc/critic.hip,
critic_tail_pgrad16_kernel<TD>): builds critic.hip with -DCRITIC_STAMPS into a temporary library (hipcc is on the GPU box),
runs flexnet_critic_td_backward at the update batch and prints the s_memtime differences (shader cycles) of block 0's first
wavefront, summed over its tiles.  The stamps drain the LDS queue at every boundary: a coarse view, not a timing."""
import ctypes as C
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import safe_marl_amd  # noqa: F401
from safe_marl_amd import _lib

tmp = tempfile.mkdtemp()
so = os.path.join(tmp, "libcritic_stamps.so")
csrc = os.path.join(ROOT, "safe-marl_amd", "csrc")
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC", "-DCRITIC_STAMPS",
                "-I" + os.path.join(ROOT, "include"), "-I" + csrc, "-o", so,
                os.path.join(csrc, "critic.hip"), os.path.join(csrc, "tdloss.hip")], check=True)
stamped = C.CDLL(so)
stamped.flexnet_critic_td_backward.argtypes = [C.POINTER(_lib.FlexCriticTailArgs), C.POINTER(_lib.FlexTdLossArgs), C.c_void_p]
real = _lib.load()


class _Shim:
    def __getattr__(self, name):
        return getattr(stamped if name == "flexnet_critic_td_backward" else real, name)


_lib.load = lambda: _Shim()
import bench


def timed(fn, n=50):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


r = bench.critic_backward_roofline(timed)
t = (C.c_ulonglong * 128)()
stamped.flexnet_debug_critic_stamps(t)
names = ["staging + barrier", "first-layer row, LayerNorm, a1, transpose store", "fc2 forward chain (64 MFMAs / tile)",
         "bias, q, TD error, dz2, transpose store", "da1 chain (64 MFMAs / tile)", "ReLU / LayerNorm backward, dz1 store",
         "phase B: dW2 (64 MFMAs / tile)", "sums after the last tile", "wait for the block's other wavefronts", "end-of-kernel fold"]
tot = [sum(t[16 * w + k] for k in range(10)) for w in range(8)]
print(f"stamped build: {r['launch_us']:.1f} us for the launches; block 0, cycles per wavefront (0-7), summed over its tiles; total {tot}")
for k, nm in enumerate(names):
    print(f"  {nm:48s} " + " ".join(f"{t[16 * w + k]:7d}" for w in range(8)) + f"   {100.0 * t[k] / max(tot[0], 1):5.1f} % of wavefront 0")
print("MFMA issue bound per chain: 5 tiles x 64 x 32 = 10240 cycles")
