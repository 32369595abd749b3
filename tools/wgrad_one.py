"""A few eager calls of csrc/wgrad.hip on the critic's first-layer shape (for a counter pass: tools/pmc_any.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from safe_marl_amd.nets import tall_wgrad
k, m, n = 32768, 64, 720
dy = torch.randn(k, m, device="cuda"); x = torch.randn(k, n, device="cuda")
for _ in range(6):
    tall_wgrad(dy, x)
torch.cuda.synchronize()
