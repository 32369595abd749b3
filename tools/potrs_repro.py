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
resulting steps pass the KKT certificate of tests/test_opf_cpu.py and the SLSQP comparison of tests/test_opf_gpu.py, so
the operands are well-formed.  THE CAUSE IS UNCONFIRMED: the faulting line itself never reached the history — the first
committed version of opf.py already used the trsm pair — and neither the run log, a fault address nor a driver fault
record of that session was kept.  "The potrsBatched route" is an inference from which call was replaced, not a finding.
If the fault is ever pursued, start from a recorded run (dmesg / the driver's fault record), not from this file.

This script only PRINTS the operands' shapes, strides and versions and checks the trsm route; it does not make the
cholesky_solve call (never re-trigger a GPU memory fault on a shared host).
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
    print(__doc__)
    if not torch.cuda.is_available():
        sys.exit(0)
    N, rhs = operands()
    L, info = torch.linalg.cholesky_ex(N)
    print("L", tuple(L.shape), L.stride(), L.dtype, "rhs", tuple(rhs.shape), rhs.stride(), "info max", int(info.max()))
    y = torch.linalg.solve_triangular(L, rhs, upper=False)
    x_trsm = torch.linalg.solve_triangular(L.transpose(-1, -2), y, upper=True)
    torch.cuda.synchronize()
    print("trsm route residual", float((N @ x_trsm - rhs).abs().max()))
    print("the call on record — NOT made here: torch.cholesky_solve(rhs, L)")
