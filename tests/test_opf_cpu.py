"""CPU coverage of the OPF comparator: the oracle against the reference's own expressions, and the
batched interior-point QP (pure torch, runs on the CPU) against SciPy."""
import numpy as np
import pytest
import torch

from oracle import opf_oracle as oo
from safe_marl_amd.network import create_network
from safe_marl_amd.series import make_synthetic_series


def _instance(net, T, day=3, first=40, n_days=6):
    s = make_synthetic_series(net, n_days=n_days)
    rows = np.asarray(s.table)[96 * day + first:96 * day + first + T]
    return rows[:, 71], rows[:, :33], rows[:, 33:66], rows[:, 66:71], np.full(5, 0.0125)


def test_oracle_solution_zeroes_the_reference_expressions():
    net = create_network()
    price, pd, qd, ppv, e0 = _instance(net, 2)
    x, f, info = oo.solve_reduced(net, {}, price, pd, qd, ppv, e0)
    assert info["success"]
    sol = oo.solution_dict(net, {}, pd, qd, ppv, e0, x)
    res = oo.opf_residuals(net, {}, pd, qd, ppv, e0, sol)
    for k in ("active", "reactive", "vdrop", "current_def", "energy", "slack_v"):
        assert res[k] < 1e-10, (k, res[k])                     # equalities opf.py:96-129,139-148
    for k in ("current_lim", "v_lim", "qpv_lim", "e_lim", "box"):
        assert res[k] < 1e-9, (k, res[k])                      # inequalities opf.py:54-65,118-137
    assert res["simultaneous"] < 1e-9                          # relaxed binaries opf.py:150-156 are tight
    assert abs(oo.opf_objective(net, {}, price, sol) - f) < 1e-12
    assert oo.first_order_gap(net, {}, price, pd, qd, ppv, e0, x) < 1e-9
    # better than the separable guess (flexibility at its unconstrained optimum, nothing else)
    x0 = np.zeros_like(x)
    x0[:, 0] = np.minimum(price[:, None] / 0.3, pd[:, [4, 9, 14, 19, 24]] * 0.5)
    f0 = oo.opf_objective(net, {}, price, oo.solution_dict(net, {}, pd, qd, ppv, e0, x0))
    assert f >= f0 - 1e-12


def test_first_period_storage_never_reaches_the_energy_balance():
    """opf.py:140-142: E[k,1] = E_init whatever Pesc[k,1], Pesd[k,1] are."""
    net = create_network()
    price, pd, qd, ppv, e0 = _instance(net, 2)
    P = oo.ReducedOPF(net, {}, price, pd, qd, ppv, e0)
    x = np.zeros((2, 4, 5))
    x[0, 3] = 0.005
    x[1, 2] = 0.004
    e = P.energy(x.ravel())
    assert np.allclose(e[0], e0) and np.allclose(e[1], e0 + 0.25 * 0.9 * 0.004)


def test_batched_interior_point_matches_scipy():
    from scipy.optimize import minimize
    from safe_marl_amd.opf import _Identity, _PeriodBlocks, _Shared, qp_ipm

    rng = np.random.default_rng(5)
    B, T, w, R, mE = 3, 4, 6, 5, 7
    n = T * w
    A = rng.normal(size=(B, T, w, w))
    Q = A @ A.transpose(0, 1, 3, 2) * 0.1                      # PSD blocks, some nearly singular
    Q[:, :, 0, :] = 0
    Q[:, :, :, 0] = 0                                          # an LP-like direction
    c = rng.normal(size=(B, n))
    lo, hi = -np.ones((B, n)), np.ones((B, n))
    lo[:, 3] = hi[:, 3] = 0.25                                 # a pinned variable
    J = rng.normal(size=(B, T, R, w))
    ju = np.abs(rng.normal(size=(B, T * R))) + 0.2
    C = rng.normal(size=(mE, n))
    cl, cu = -np.abs(rng.normal(size=(B, mE))) - 0.1, np.abs(rng.normal(size=(B, mE))) + 0.1
    t = lambda a: torch.tensor(a, dtype=torch.float64)
    free = t(hi - lo) >= 1e-9
    pin = (~free).double()
    blocks = [(_Identity(), t(lo) - pin, t(hi) + pin), (_PeriodBlocks(t(J)), None, t(ju)), (_Shared(t(C)), t(cl), t(cu))]
    x0 = torch.where(free, t(0.5 * (lo + hi)), t(lo))
    x, info = qp_ipm(t(Q), t(c), blocks, x0, free=free)
    assert bool(info["converged"].all())
    # (1) KKT certificate from the solver's own multipliers: sufficient for a convex QP, independent of any other solver.
    # duals come per one-sided row set in the order (upper, lower) of each block
    z = [d.numpy() for d in info["duals"]]
    xs = x.numpy()
    for b in range(B):
        Qd = np.zeros((n, n))
        Jd = np.zeros((T * R, n))
        for k in range(T):
            Qd[k * w:(k + 1) * w, k * w:(k + 1) * w] = Q[b, k]
            Jd[k * R:(k + 1) * R, k * w:(k + 1) * w] = J[b, k]
        xb = xs[b]
        fr = (hi[b] - lo[b]) >= 1e-9
        rows = [(np.eye(n), (hi[b] + ~fr), +1), (np.eye(n), (lo[b] - ~fr), -1), (Jd, ju[b], +1), (C, cu[b], +1), (C, cl[b], -1)]
        grad = Qd @ xb + c[b]
        for (Am, bound, sg), zz in zip(rows, z):
            zb = zz[b]
            assert zb.min() >= 0
            slack = sg * (bound - Am @ xb)
            assert slack.min() > -1e-8                                  # primal feasibility
            assert np.abs(zb * slack).max() < 1e-7                      # complementarity
            grad = grad + sg * (Am.T @ zb)
        assert np.abs(grad[fr]).max() < 2e-5 * max(1.0, np.abs(c[b]).max())   # stationarity on the free variables (the dual residual floors with the conditioning)
        assert abs(xb[3] - 0.25) < 1e-12                                # the pinned one never moved
        # (2) and SciPy agrees on the optimal value
        from scipy.optimize import Bounds, LinearConstraint
        ref = minimize(lambda v: 0.5 * v @ Qd @ v + c[b] @ v, np.clip(np.zeros(n), lo[b], hi[b]), jac=lambda v: Qd @ v + c[b],
                       hess=lambda v: Qd, bounds=Bounds(lo[b], hi[b]), method="trust-constr",
                       constraints=[LinearConstraint(Jd, -np.inf, ju[b]), LinearConstraint(C, cl[b], cu[b])],
                       options=dict(gtol=1e-10, xtol=1e-12, maxiter=3000))
        f_ipm = 0.5 * xb @ Qd @ xb + c[b] @ xb
        assert f_ipm <= ref.fun + 1e-6 * max(1.0, abs(ref.fun))
        assert abs(f_ipm - ref.fun) < 1e-4 * max(1.0, abs(ref.fun))


def test_energy_chain_operator_equals_its_matrix():
    """opf.py:139-148 as cumulative sums vs the explicit [T*na, T*4*na] matrix."""
    from safe_marl_amd.opf import BatchedOPF, _EnergyChain, _Shared
    net = create_network()
    opf = BatchedOPF(net, device="cpu")
    T, B = 6, 2
    C = opf._energy_matrix(T, torch.float64)
    op, ref = _EnergyChain(T, opf.na, opf.dt * 0.9, opf.dt / 0.9), _Shared(C)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, T * opf.w, dtype=torch.float64, generator=g)
    y = torch.randn(B, T * opf.na, dtype=torch.float64, generator=g)
    d = torch.rand(B, T * opf.na, dtype=torch.float64, generator=g)
    assert torch.allclose(op.apply(x), ref.apply(x), atol=1e-14)
    assert torch.allclose(op.apply_t(y), ref.apply_t(y), atol=1e-14)
    n = T * opf.w
    N1, N2 = torch.zeros(B, n, n, dtype=torch.float64), torch.zeros(B, n, n, dtype=torch.float64)
    op.add_gram(N1, d)
    ref.add_gram(N2, d)
    assert torch.allclose(N1, N2, atol=1e-14)
    # and it is the chain: E = e0 + C x
    xx = torch.rand(B, T, 4, opf.na, dtype=torch.float64, generator=g)
    e0 = torch.full((B, opf.na), 0.0125, dtype=torch.float64)
    assert torch.allclose(opf.energy(e0, xx), e0[:, None, :] + op.apply(xx.reshape(B, -1)).view(B, T, opf.na), atol=1e-15)


def test_block_elimination_equals_the_dense_normal_equations():
    """qp_ipm's structured factorisation (per-period elimination of the controls the energy chain does not touch, Schur
    complement on the storage controls) solves the same system as the dense one, pinned variables included."""
    from safe_marl_amd import opf
    torch.manual_seed(0)
    B, T, na = 2, 5, 3
    w = 4 * na
    n = T * w
    A = torch.randn(B, T, w, w, dtype=torch.float64)
    Q = A @ A.transpose(2, 3) * 0.1
    J = torch.randn(B, T, 7, w, dtype=torch.float64)
    chain = opf._EnergyChain(T, na, 0.2, 0.3)
    sets = [(opf._Identity(), 1.0, None), (opf._Identity(), -1.0, None), (opf._PeriodBlocks(J), 1.0, None),
            (chain, 1.0, None), (chain, -1.0, None)]
    sizes = (n, n, T * 7, T * na, T * na)
    s = [torch.rand(B, m, dtype=torch.float64) + 0.1 for m in sizes]
    z = [torch.rand(B, m, dtype=torch.float64) * 10 ** torch.randint(-6, 3, (B, m)).double() + 1e-9 for m in sizes]
    fm = torch.ones(B, n, dtype=torch.float64)
    fm[:, 5] = 0
    fm[:, 2 * na + 1] = 0                                        # a pinned local and a pinned storage control
    rhs = torch.randn(B, n, dtype=torch.float64) * fm
    x1 = opf._factor_dense(Q, sets, s, z, fm, 1e-12)(rhs)
    x2 = opf._factor_structured(Q, sets, s, z, fm, 1e-12)(rhs)
    assert (x1 - x2).abs().max().item() < 1e-9 * max(1.0, x1.abs().max().item())
    assert x2[:, 5].abs().max().item() == 0.0


def _chain_qp(B, T, na, R, seed):
    rng = np.random.default_rng(seed)
    w, n = 4 * na, T * 4 * na
    A = rng.normal(size=(B, T, w, w))
    Q = A @ A.transpose(0, 1, 3, 2) * 0.05
    Q[:, :, 0, :] = 0
    Q[:, :, :, 0] = 0
    lo, hi = -np.ones((B, T, w)), np.ones((B, T, w))
    lo[:, ::2, na + 1] = hi[:, ::2, na + 1] = 0.0
    t = lambda a: torch.tensor(a, dtype=torch.float64)
    return dict(Q=t(Q), c=t(rng.normal(size=(B, n))), lo=t(lo.reshape(B, n)), hi=t(hi.reshape(B, n)),
                jv=t(rng.normal(size=(B, T, R, w))), ji=t(rng.normal(size=(B, T, R, w))),
                v_hi=t(np.abs(rng.normal(size=(B, T * R))) + 0.2), v_lo=t(-np.abs(rng.normal(size=(B, T * R))) - 0.2),
                i_hi=t(np.abs(rng.normal(size=(B, T * R))) + 0.2), e_hi=t(np.abs(rng.normal(size=(B, T * na))) * 0.3 + 0.05),
                e_lo=t(-np.abs(rng.normal(size=(B, T * na))) * 0.3 - 0.05))


def test_riccati_recursion_solves_the_newton_system_of_the_dense_factorisation():
    """The algorithm of csrc/opf.hip (include/flexopf.h) in torch: its Newton step equals the dense Cholesky one while z / s
    spans twelve decades, and the interior-point iteration on top of it ends where the dense one ends, in as many steps."""
    from safe_marl_amd.opf import _EnergyChain, _Identity, _PeriodBlocks, _factor_dense, _factor_riccati, qp_ipm
    B, T, na, R = 2, 9, 5, 6
    p = _chain_qp(B, T, na, R, 11)
    w, n = 4 * na, T * 4 * na
    chain = _EnergyChain(T, na, 0.25 * 0.9, 0.25 / 0.9)
    sets = [(_Identity(), 1.0, None), (_Identity(), -1.0, None), (_PeriodBlocks(p["jv"]), 1.0, None), (_PeriodBlocks(p["jv"]), -1.0, None),
            (_PeriodBlocks(p["ji"]), 1.0, None), (chain, 1.0, None), (chain, -1.0, None)]
    g = torch.Generator().manual_seed(4)
    sizes = [n, n, T * R, T * R, T * R, T * na, T * na]
    s = [torch.rand(B, m, dtype=torch.float64, generator=g) + 0.1 for m in sizes]
    z = [torch.rand(B, m, dtype=torch.float64, generator=g) * 10.0 ** torch.randint(-6, 6, (B, m), generator=g).double() for m in sizes]
    fmask = ((p["hi"] - p["lo"]) >= 1e-9).double()
    rhs = torch.randn(B, n, dtype=torch.float64, generator=g) * fmask
    xd = _factor_dense(p["Q"], sets, s, z, fmask, 1e-12)(rhs) * fmask
    xr = _factor_riccati(p["Q"], sets, s, z, fmask, 1e-12)(rhs) * fmask
    assert (xd - xr).abs().max().item() < 1e-8 * max(1.0, xd.abs().max().item())
    free = fmask > 0.5
    pin = (~free).double()
    blocks = [(_Identity(), p["lo"] - pin, p["hi"] + pin), (_PeriodBlocks(p["jv"]), p["v_lo"], p["v_hi"]),
              (_PeriodBlocks(p["ji"]), None, p["i_hi"]), (chain, p["e_lo"], p["e_hi"])]
    x0 = torch.where(free, 0.5 * (p["lo"] + p["hi"]), p["lo"])
    xa, ia = qp_ipm(p["Q"], p["c"], blocks, x0, free=free)
    xb, ib = qp_ipm(p["Q"], p["c"], blocks, x0, free=free, factor="riccati")
    assert bool(ia["converged"].all()) and bool(ib["converged"].all())
    assert ib["iters"] <= ia["iters"] + 1
    f = lambda x: 0.5 * torch.einsum("btv,btvw,btw->b", x.view(B, T, w), p["Q"], x.view(B, T, w)) + (p["c"] * x).sum(1)
    assert (f(xa) - f(xb)).abs().max().item() < 1e-8
    assert (xa - xb).abs().max().item() < 2e-5
