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
