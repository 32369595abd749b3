"""How long do the OPF interior point's dense factorisation and solves take with the library path (torch.linalg on
[B, 960, 960] fp64)?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import safe_marl_amd
from safe_marl_amd import opf
B, n = int(sys.argv[1]) if len(sys.argv) > 1 else 128, 960
g = torch.Generator(device="cpu").manual_seed(0)
a = torch.randn(B, n, n, dtype=torch.float64, generator=g)
S = (a @ a.transpose(1, 2) / n + torch.eye(n, dtype=torch.float64)).cuda()
rhs = torch.randn(B, n, 1, dtype=torch.float64, generator=g).cuda()
def t(f, reps=5):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
L = opf._chol_retry(S.clone(), "x")
print("B", B, "cholesky_ex+retry check: %.2f ms" % t(lambda: opf._chol_retry(S.clone(), "x")))
print("tri_solve pair: %.2f ms" % t(lambda: opf._tri_solve(L, rhs)))
x = opf._tri_solve(L, rhs)
print("residual", float((S @ x - rhs).abs().max()))
if hasattr(opf, "blocked_cholesky"):
    F = opf.blocked_cholesky(S.clone())
    print("blocked factor: %.2f ms" % t(lambda: opf.blocked_cholesky(S.clone())))
    print("blocked solve: %.2f ms" % t(lambda: F.solve(rhs)))
    x2 = F.solve(rhs)
    print("blocked residual", float((S @ x2 - rhs).abs().max()), "vs library", float((x2 - x).abs().max()))
