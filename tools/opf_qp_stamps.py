"""Where an interior-point iteration of csrc/opf.hip spends its time: cycles per phase of work-group 0 from a DIAGNOSTIC build
(-DQP_STAMPS: wall-clock stamps by thread 0, printed by the kernel).  Build it first (repo root):
    hipcc -O3 --offload-arch=gfx950 -std=c++17 -shared -fPIC -DQP_STAMPS -Iinclude -Isafe-marl_amd/csrc \
        -o tools/variants/libflexenv_hip_qpstamps.so safe-marl_amd/csrc/*.hip
then   python tools/opf_qp_stamps.py [B]      (one OPF solve of B days x 96 periods; never quote its run time)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import safe_marl_amd  # noqa: F401
from safe_marl_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "tools", "variants", "libflexenv_hip_qpstamps.so")
from safe_marl_amd.network import create_network
from safe_marl_amd.series import make_synthetic_series
from safe_marl_amd.opf import BatchedOPF

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
net = create_network()
tab = np.asarray(make_synthetic_series(net, n_days=B + 8).table)
rows = np.stack([tab[96 * (3 + b):96 * (3 + b) + 96] for b in range(B)])
r = BatchedOPF(net).solve(rows[:, :, 71], rows[:, :, :33], rows[:, :, 33:66], rows[:, :, 66:71], np.full((B, 5), 0.0125))
torch.cuda.synchronize()
print("outer iterations", r["outer_iters"], "objective mean", r["objective"].mean().item())
