"""Native QP (csrc/opf.hip) against the torch iteration after k iterations each: where they part.  python tools/opf_qp_debug.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import safe_marl_amd  # noqa: F401
from safe_marl_amd.opf import _EnergyChain, _Identity, _PeriodBlocks, qp_ipm, qp_ipm_native
from test_opf_native_gpu import _random_qp

dev = torch.device("cuda:0")
B, T, na, R = (int(v) for v in (sys.argv[1:5] if len(sys.argv) >= 5 else (2, 7, 5, 6)))
p = _random_qp(B, T, na, R, 1)
t = lambda a: torch.tensor(a, dtype=torch.float64, device=dev)
lo, hi = t(p["lo"]), t(p["hi"])
free = (hi - lo) >= 1e-9
pin = (~free).double()
x0 = torch.where(free, 0.5 * (lo + hi), lo)
jv, ji = t(p["jv"]), t(p["ji"])
blocks = [(_Identity(), lo - pin, hi + pin), (_PeriodBlocks(jv), t(p["v_lo"]), t(p["v_hi"])),
          (_PeriodBlocks(ji), None, t(p["i_hi"])), (_EnergyChain(T, na, p["a"], p["b"]), t(p["e_lo"]), t(p["e_hi"]))]
for k in (1, 2, 3, 5, 10, 20, 40, 80):
    try:
        xr, ir = qp_ipm(t(p["Q"]), t(p["c"]), blocks, x0, free=free, max_iter=k)
    except RuntimeError as e:
        xr, ir = None, None
    try:
        xn, inn = qp_ipm_native(t(p["Q"]), t(p["c"]), lo - pin, hi + pin, free, jv, t(p["v_lo"]), t(p["v_hi"]), ji, t(p["i_hi"]),
                                p["a"], p["b"], t(p["e_lo"]), t(p["e_hi"]), x0, max_iter=k)
    except RuntimeError as e:
        print(k, "native raised", e)
        continue
    if xr is None:
        print(k, "torch raised; native mu", inn["mu"].tolist())
        continue
    dz = max((a - b).abs().max().item() for a, b in zip(ir["duals"], inn["duals"]))
    print(f"iters {k:3d}: |dx| {(xr - xn).abs().max().item():.3e}  |dz| {dz:.3e}  mu torch {ir['mu'].max().item():.3e} native {inn['mu'].max().item():.3e}"
          f"  res_p {ir['res_p'].max().item():.2e} / {inn['res_p'].max().item():.2e}  res_d {ir['res_d'].max().item():.2e} / {inn['res_d'].max().item():.2e}"
          f"  iters {ir['iters']} / {inn['iters']}  conv {ir['converged'].tolist()} / {inn['converged'].tolist()} floored {inn['floored_pivots'].tolist()}")
