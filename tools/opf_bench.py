"""Timing of the batched OPF comparator (safe-marl_amd/opf.py) on one MI355X: whole days (T = 96) per second for
several batch sizes, split into the power-flow launches (state + central-difference sensitivities) and the
interior-point iterations, with the CPU oracle (SciPy SLSQP, reduced space, T = 4) timed beside it for context.
    python tools/opf_bench.py [B ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import safe_marl_amd  # noqa: F401
from safe_marl_amd.network import create_network
from safe_marl_amd.series import make_synthetic_series
from safe_marl_amd import opf as opf_mod
from safe_marl_amd.opf import BatchedOPF

net = create_network()
series = make_synthetic_series(net, n_days=400)
tab = np.asarray(series.table)
batches = [int(a) for a in sys.argv[1:]] or [1, 32, 128]
T = 96
opf = BatchedOPF(net)
# time the two phases by wrapping them
acc = dict(pf=0.0, ipm=0.0, pf_solves=0, ipm_iters=0)
_lin, _ipm, _ipm_native = BatchedOPF.linearise, opf_mod.qp_ipm, opf_mod.qp_ipm_native


def lin(self, pd, qd, ppv, x, **kw):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = _lin(self, pd, qd, ppv, x, **kw)
    torch.cuda.synchronize(); acc["pf"] += time.perf_counter() - t0
    acc["pf_solves"] += x.shape[0] * x.shape[1] * (1 + 2 * self.w)
    return r


def ipm(*a, **kw):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    x, info = _ipm(*a, **kw)
    torch.cuda.synchronize(); acc["ipm"] += time.perf_counter() - t0
    acc["ipm_iters"] += info["iters"]
    return x, info


def ipm_native(*a, **kw):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    x, info = _ipm_native(*a, **kw)
    torch.cuda.synchronize(); acc["ipm"] += time.perf_counter() - t0
    acc["ipm_iters"] += info["iters"]
    acc["qp_solves"] = acc.get("qp_solves", 0) + 1
    return x, info


BatchedOPF.linearise = lin
opf_mod.qp_ipm = ipm
opf_mod.qp_ipm_native = ipm_native
if os.environ.get("FLEX_OPF_TORCH_QP") == "1":
    opf_mod.NATIVE_QP = False
for B in batches:
    rows = np.stack([tab[96 * (3 + b):96 * (3 + b) + T] for b in range(B)])
    args = (rows[:, :, 71], rows[:, :, :33], rows[:, :, 33:66], rows[:, :, 66:71], np.full((B, 5), 0.0125))
    if B == batches[0]:
        opf.solve(*[a[:1] for a in args])                      # warm-up (library load, rocBLAS/rocSOLVER handles)
    for k in acc:
        acc[k] = 0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = opf.solve(*args)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"B={B:4d} days x T=96: {dt:7.2f} s  ({B / dt:7.2f} days/s)  outer {r['outer_iters']:2d}  ipm iters {acc['ipm_iters']:3d}  "
          f"power flow {acc['pf']:.3f} s for {acc['pf_solves']:,} solves ({acc['pf_solves'] / max(acc['pf'], 1e-9) / 1e6:.1f} M/s)  "
          f"interior point {acc['ipm']:.2f} s ({1e3 * acc['ipm'] / max(acc['ipm_iters'], 1):.1f} ms/iter)  "
          f"objective mean {r['objective'].mean().item():+.6f}  min V {r['Vsqr'].min().sqrt().item():.4f}", flush=True)

from oracle import opf_oracle as oo
rows = tab[96 * 3 + 40:96 * 3 + 44]
t0 = time.perf_counter()
x, f, info = oo.solve_reduced(net, {}, rows[:, 71], rows[:, :33], rows[:, 33:66], rows[:, 66:71], np.full(5, 0.0125))
print(f"CPU oracle (SciPy SLSQP over the controls, NumPy power flow), T=4, one instance: {time.perf_counter() - t0:.2f} s, "
      f"{info['nit']} iterations")
