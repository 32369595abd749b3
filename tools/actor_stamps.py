"""Phase breakdown of one wavefront of the matrix-core actor kernel: builds csrc/actor.hip with
-DACTOR_STAMPS into a temporary library (hipcc is on the GPU box), runs it on a one-tile and a rollout-sized batch and
prints the s_memtime differences (core cycles) between the stamps of block 0's first wavefront."""
import ctypes as C
import os
import subprocess
import sys
import tempfile
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import safe_marl_amd  # noqa: F401
from safe_marl_amd import _lib
from safe_marl_amd.nets import RNNAgent, fused_actor_forward

tmp = tempfile.mkdtemp()
so = os.path.join(tmp, "libactor_stamps.so")
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC", "-DACTOR_STAMPS",
                "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "safe-marl_amd", "csrc"), "-o", so,
                os.path.join(ROOT, "safe-marl_amd", "csrc", "actor.hip")], check=True)
stamped = C.CDLL(so)
stamped.flexnet_actor_forward.argtypes = [C.POINTER(_lib.FlexActorArgs), C.c_void_p]
real = _lib.load()


class _Shim:                     # the stamped actor entry point, everything else from the product library
    def __getattr__(self, name):
        return getattr(stamped if name == "flexnet_actor_forward" else real, name)


_lib.load = lambda: _Shim()
args = types.SimpleNamespace(hid_size=64, layernorm=True, action_dim=4, agent_num=5, hid_activation="relu")
agent = RNNAgent(149, args).cuda()
for b in (5, 4096):
    obs = torch.randn(b, 5, 144, device="cuda")
    hid = torch.randn(b, 5, 64, device="cuda")
    with torch.no_grad():
        for _ in range(3):
            fused_actor_forward(agent, obs, hid, 5, True)
    torch.cuda.synchronize()
    t = (C.c_ulonglong * 16)()
    stamped.flexnet_debug_actor_stamps(t)
    print("%6d rows, wavefront 0 (one whole tile): weight staging %6d  fc1 %6d  LayerNorm %5d  GRU %6d  fc2 + stores %5d   cycles" %
          (b * 5, t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3], t[5] - t[4]))
    if t[13] > t[8]:
        print("%6d rows, wavefront 4 (a quarter of the fifth tile): staging %6d  fc1 + rendezvous %6d  LayerNorm %5d  GRU + rendezvous %6d  "
              "fc2 + stores %5d   cycles" % (b * 5, t[9] - t[8], t[10] - t[9], t[11] - t[10], t[12] - t[11], t[13] - t[12]))
for b in (32768,):                                   # the 32-row kernel (update batches)
    obs = torch.randn(b, 5, 144, device="cuda")
    hid = torch.randn(b, 5, 64, device="cuda")
    with torch.no_grad():
        for _ in range(3):
            fused_actor_forward(agent, obs, hid, 5, True)
    torch.cuda.synchronize()
    t = (C.c_ulonglong * 16)()
    stamped.flexnet_debug_actor_stamps(t)
    print("%6d rows (32-row kernel, last tile of block 0's first wavefront): weight staging %6d  fc1 %6d  LayerNorm %5d  GRU %6d  "
          "fc2 + stores %5d   cycles" % (b * 5, t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3], t[5] - t[4]))
