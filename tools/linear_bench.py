"""us per launch of the critic's first layer at the update batch: flexnet_linear2 (csrc/linear.hip) against the two library GEMMs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import safe_marl_amd
from safe_marl_amd import nets
def timed(fn, n=200):
    for _ in range(20): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for rows in (32768, 36864, 131072):
    W = torch.randn(64, 745, device="cuda") * 0.3; b = torch.randn(64, device="cuda")
    x1 = torch.randn(rows, 720, device="cuda"); x2 = torch.randn(rows, 20, device="cuda")
    with torch.no_grad():
        t_f = timed(lambda: nets.critic_first_layer(b, x1, x2, W, 725))
        nets.CRITIC_FC1_FUSED = False
        t_l = timed(lambda: nets.critic_first_layer(b, x1, x2, W, 725))
        nets.CRITIC_FC1_FUSED = True
    fl = 2.0 * rows * 740 * 64
    print(f"rows {rows}: linear2 {t_f:.1f} us ({fl / t_f / 1e6:.1f} TFLOP/s, {fl / t_f / 1e6 / 157.3:.2f} of the fp32 MFMA peak), library pair {t_l:.1f} us ({fl / t_l / 1e6:.1f} TFLOP/s)")
