#!/usr/bin/env python3
"""Minimal statement of the round-1 fault on record: NOT part of any test and NOT run by the build —
a GPU memory fault can take the whole host down, so this file documents the call, it does not exercise it.

What faulted (round 1, ROCm 7.2.0 image, torch 2.10.0+rocm7.0, MI355X): the solve step of the OPF interior point written as

    L, info = torch.linalg.cholesky_ex(N)            # N: [32, 1920, 1920] fp64, contiguous, SPD  -> fine
    x = torch.cholesky_solve(rhs, L)                 # rhs: [32, 1920, 1] fp64, contiguous        -> GPU memory fault

For batch > 1 and a single right-hand side ATen routes cholesky_solve to hipSOLVER's potrsBatched (array-of-pointers
interface) after cloning L into column-major batches (lda = n); both operands here are freshly allocated tensors with
standard strides — L.stride() = (1920*1920, 1920, 1), rhs.stride() = (1920, 1, 1), storage offsets 0, the last matrix
starts 31 * 1920^2 = 114 278 400 elements (914 MB) into the allocation, below 2^31 in elements and in bytes.  The same
L and rhs go through two rocBLAS strided-batched trsm calls (torch.linalg.solve_triangular) without trouble, and the
resulting steps pass the KKT certificate of tests/test_opf_cpu.py and the SLSQP comparison of tests/test_opf_gpu.py: the
operands are well-formed, the fault is specific to the potrsBatched route.  (The faulting line itself never reached the
history — the first committed version of opf.py already used the trsm pair — and the run log of that session was not
kept, so the fault address is not on record; the shapes, strides and versions above are.)

Usage (at your own risk, on a box you may lose):  python tools/potrs_repro.py --run
"""
import sys

import torch


def operands(batch=32, n=1920, device="cuda"):
    g = torch.Generator(device="cpu").manual_seed(0)
    a = torch.randn(batch, n, n, dtype=torch.float64, generator=g)
    spd = (a @ a.transpose(1, 2) / n + torch.eye(n, dtype=torch.float64)).to(device)
    rhs = torch.randn(batch, n, 1, dtype=torch.float64, generator=g).to(device)
    return spd, rhs


if __name__ == "__main__":
    print("torch", torch.__version__, "hip", torch.version.hip)
    if "--run" not in sys.argv:
        print(__doc__)
        sys.exit(0)
    N, rhs = operands()
    L, info = torch.linalg.cholesky_ex(N)
    print("L", tuple(L.shape), L.stride(), L.dtype, "rhs", tuple(rhs.shape), rhs.stride(), "info max", int(info.max()))
    y = torch.linalg.solve_triangular(L, rhs, upper=False)
    x_trsm = torch.linalg.solve_triangular(L.transpose(-1, -2), y, upper=True)
    torch.cuda.synchronize()
    print("trsm route residual", float((N @ x_trsm - rhs).abs().max()))
    x = torch.cholesky_solve(rhs, L)                 # the call on record
    torch.cuda.synchronize()
    print("potrs route residual", float((N @ x - rhs).abs().max()))
