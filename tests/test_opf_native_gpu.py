"""GPU: the OPF comparator's QP through the HIP entry point (include/flexopf.h, csrc/opf.hip: whole interior-point iteration
in one persistent work-group per day, Riccati recursion over the periods) against the pure-torch iteration it restates
(safe-marl_amd/opf.py: qp_ipm, dense / block-eliminated factorisations), and the sequential convex programme on top of it
against the same programme on the torch QP.  Reference: utils/opf.py:13-192."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _random_qp(B, T, na, R, seed, pin=True):
    rng = np.random.default_rng(seed)
    w, n = 4 * na, T * 4 * na
    A = rng.normal(size=(B, T, w, w))
    Q = A @ A.transpose(0, 1, 3, 2) * 0.05
    Q[:, :, 0, :] = 0
    Q[:, :, :, 0] = 0                                          # an LP-like direction
    c = rng.normal(size=(B, n))
    lo, hi = -np.ones((B, T, w)), np.ones((B, T, w))
    if pin:
        lo[:, ::2, na + 1] = hi[:, ::2, na + 1] = 0.0          # a reactive-power control pinned every other period
    jv, ji = rng.normal(size=(B, T, R, w)), rng.normal(size=(B, T, R, w))
    v_hi = np.abs(rng.normal(size=(B, T * R))) + 0.2
    v_lo = -np.abs(rng.normal(size=(B, T * R))) - 0.2
    i_hi = np.abs(rng.normal(size=(B, T * R))) + 0.2
    e_hi = np.abs(rng.normal(size=(B, T * na))) * 0.3 + 0.05
    e_lo = -np.abs(rng.normal(size=(B, T * na))) * 0.3 - 0.05
    return dict(Q=Q, c=c, lo=lo.reshape(B, n), hi=hi.reshape(B, n), jv=jv, ji=ji, v_lo=v_lo, v_hi=v_hi, i_hi=i_hi, e_lo=e_lo, e_hi=e_hi,
                a=0.25 * 0.9, b=0.25 / 0.9)


def _both(p, dev):
    from safe_marl_amd.opf import _EnergyChain, _Identity, _PeriodBlocks, qp_ipm, qp_ipm_native
    t = lambda a: torch.tensor(a, dtype=torch.float64, device=dev)
    B, T, w, _ = p["Q"].shape
    na = w // 4
    lo, hi = t(p["lo"]), t(p["hi"])
    free = (hi - lo) >= 1e-9
    pin = (~free).double()
    x0 = torch.where(free, 0.5 * (lo + hi), lo)
    jv, ji = t(p["jv"]), t(p["ji"])
    blocks = [(_Identity(), lo - pin, hi + pin), (_PeriodBlocks(jv), t(p["v_lo"]), t(p["v_hi"])),
              (_PeriodBlocks(ji), None, t(p["i_hi"])), (_EnergyChain(T, na, p["a"], p["b"]), t(p["e_lo"]), t(p["e_hi"]))]
    xr, ir = qp_ipm(t(p["Q"]), t(p["c"]), blocks, x0, free=free)
    xn, inn = qp_ipm_native(t(p["Q"]), t(p["c"]), lo - pin, hi + pin, free, jv, t(p["v_lo"]), t(p["v_hi"]), ji, t(p["i_hi"]),
                            p["a"], p["b"], t(p["e_lo"]), t(p["e_hi"]), x0)
    return xr, ir, xn, inn, blocks, free


@pytest.mark.parametrize("B,T,na,R,seed", [(3, 7, 5, 6, 1), (2, 12, 3, 9, 2), (4, 24, 5, 32, 3), (1, 5, 5, 44, 4)])
def test_native_qp_equals_the_torch_iteration(B, T, na, R, seed):
    dev = torch.device("cuda:0")
    p = _random_qp(B, T, na, R, seed)
    xr, ir, xn, inn, blocks, free = _both(p, dev)
    assert bool(ir["converged"].all()) and bool(inn["converged"].all())
    Q, c = torch.tensor(p["Q"], device=dev), torch.tensor(p["c"], device=dev)
    w = 4 * na

    def obj(x):
        xv = x.view(B, T, w)
        return 0.5 * torch.einsum("btv,btvw,btw->b", xv, Q, xv) + (c * x).sum(1)

    fr, fn = obj(xr), obj(xn)
    assert (fr - fn).abs().max().item() < 1e-8 * max(1.0, fr.abs().max().item())
    assert (xr - xn).abs().max().item() < 2e-5          # (LP-like directions: the minimiser is flat along them)
    assert (xn[~free] - xr[~free]).abs().max().item() == 0.0
    # KKT certificate from the native solve's own multipliers (sufficient for a convex QP)
    grad = torch.einsum("btvw,btw->btv", Q, xn.view(B, T, w)).reshape(B, -1) + c
    k = 0
    for rows, lower, upper in blocks:
        ax = rows.apply(xn)
        for sg, bound in ((1.0, upper), (-1.0, lower)):
            if bound is None:
                continue
            z = inn["duals"][k]
            k += 1
            assert z.min().item() >= 0.0
            slack = sg * (bound - ax)
            assert slack.min().item() > -1e-8
            assert (z * slack).abs().max().item() < 1e-7
            grad = grad + sg * rows.apply_t(z)
    assert k == 7
    assert grad[free].abs().max().item() < 1e-4 * max(1.0, c.abs().max().item())    # (floors with the conditioning, as in qp_ipm)
    assert int(inn["iters"]) <= int(ir["iters"]) + 3


def test_native_qp_rejects_sizes_it_was_not_built_for():
    import ctypes as C
    from safe_marl_amd import _lib
    lib = _lib.load()
    assert lib.flexopf_qp_work_doubles(96, 5, 32) > 0
    assert lib.flexopf_qp_work_doubles(129, 5, 32) < 0 and lib.flexopf_qp_work_doubles(96, 6, 32) < 0
    assert lib.flexopf_qp_work_doubles(96, 5, 65) < 0
    a = _lib.FlexQpArgs()
    a.batch, a.periods, a.n_agents, a.rows, a.max_iter, a.tol = 1, 96, 5, 32, 80, 1e-11
    assert lib.flexopf_qp_solve(C.byref(a), None) < 0        # null buffers


def test_opf_on_the_native_qp_equals_opf_on_the_torch_qp():
    """The sequential convex programme of opf.py with either QP solve: same controls, same objective (one day of 24 periods and
    two days of 8 under a heavier load; the binding-limit and infeasible cases are tests/test_opf_gpu.py's)."""
    from safe_marl_amd import opf as opf_mod
    from safe_marl_amd.network import create_network
    from safe_marl_amd.series import make_synthetic_series
    net = create_network()
    tab = np.asarray(make_synthetic_series(net, n_days=8).table)
    solver = opf_mod.BatchedOPF(net)
    for B, T, scale in ((1, 24, 1.0), (2, 8, 1.2)):
        rows = np.stack([tab[96 * (2 + b) + 36:96 * (2 + b) + 36 + T] for b in range(B)])
        args = (rows[:, :, 71], rows[:, :, :33] * scale, rows[:, :, 33:66] * scale, rows[:, :, 66:71], np.full((B, 5), 0.0125))
        opf_mod.NATIVE_QP = False
        try:
            ref = solver.solve(*args)
        finally:
            opf_mod.NATIVE_QP = True
        out = solver.solve(*args)
        assert (out["objective"] - ref["objective"]).abs().max().item() < 1e-9
        assert (out["x"] - ref["x"]).abs().max().item() < 5e-6
        assert out["outer_iters"] <= ref["outer_iters"] + 1


def test_more_days_than_compute_units():
    """One work-group per day, no co-operation between work-groups: a batch larger than the chip (300 days on 256 CUs) just
    queues.  Every day equals the same day solved alone in a batch of one (same launch-independent arithmetic: bit for bit)."""
    from safe_marl_amd.opf import qp_ipm_native
    dev = torch.device("cuda:0")
    B, T, na, R = 300, 3, 5, 4
    p = _random_qp(B, T, na, R, 9)
    t = lambda a: torch.tensor(a, dtype=torch.float64, device=dev)
    lo, hi = t(p["lo"]), t(p["hi"])
    free = (hi - lo) >= 1e-9
    pin = (~free).double()
    x0 = torch.where(free, 0.5 * (lo + hi), lo)
    args = [t(p["Q"]), t(p["c"]), lo - pin, hi + pin, free, t(p["jv"]), t(p["v_lo"]), t(p["v_hi"]), t(p["ji"]), t(p["i_hi"])]
    tail = [t(p["e_lo"]), t(p["e_hi"]), x0]
    x, info = qp_ipm_native(*args, p["a"], p["b"], *tail)
    assert bool(info["converged"].all()) and bool(torch.isfinite(x).all())
    for b in (0, 137, 255, 256, 299):
        xb, ib = qp_ipm_native(*[a[b:b + 1] for a in args], p["a"], p["b"], *[a[b:b + 1] for a in tail])
        assert torch.equal(xb[0], x[b])
        assert torch.equal(ib["duals"][3][0], info["duals"][3][b])


def test_horizons_beyond_the_kernels_limits_fall_back_to_the_torch_iteration():
    """BatchedOPF.solve on the GPU used to raise for programmes flexopf_qp_solve was not built for (T > 128, more
    than 64 rows, more than 5 agents); it now solves them on the torch iteration, as before the native kernel existed, and
    says so once (util.note_fallback).  T = 130 periods of a quarter hour, two days."""
    import warnings
    from safe_marl_amd import util
    from safe_marl_amd.opf import BatchedOPF
    from safe_marl_amd.network import create_network
    from safe_marl_amd.series import make_synthetic_series
    from tests.test_opf_gpu import _check_against_reference_expressions, _as_reference_solution
    net = create_network()
    s = make_synthetic_series(net, n_days=12)
    tab = np.asarray(s.table)
    T, B = 130, 2
    rows = np.stack([tab[96 * (2 + 2 * b) + 30:96 * (2 + 2 * b) + 30 + T] for b in range(B)])
    price, pd, qd, ppv, e0 = rows[:, :, 71], rows[:, :, :33], rows[:, :, 33:66], rows[:, :, 66:71], np.full((B, 5), 0.0125)
    before = util.FALLBACKS.get("opf.qp_ipm_native", 0)
    opf = BatchedOPF(net)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        r = opf.solve(price, pd, qd, ppv, e0, max_outer=6)
    assert util.FALLBACKS.get("opf.qp_ipm_native", 0) > before
    for b in range(B):
        _check_against_reference_expressions(net, price[b], pd[b], qd[b], ppv[b], e0[b], _as_reference_solution(opf, r, b),
                                             r["objective"][b].item())


def test_the_outer_iteration_contracts_on_every_day_of_a_large_batch():
    """utils/opf.py:13-192 as a sequential convex programme (opf.BatchedOPF.solve) on 128 days of 96 periods: the control move
    shrinks by about a decade per outer iteration on EVERY day — 9.9e-2, 5e-3, 4.6e-4, 4.3e-5, 3.9e-6, 4e-7 — and the batch is
    done in six.  Round 5 found one of these days held at 4.8e-5 for four iterations (ten in all, half the throughput): its
    central-difference sensitivities came from sweep solves that may stop a sweep apart at two neighbouring trial points
    (profiles/r05bm_opf_outer.txt); the sensitivities now come from the tree Newton solver."""
    from safe_marl_amd.network import create_network
    from safe_marl_amd.opf import BatchedOPF
    from safe_marl_amd.series import make_synthetic_series
    B = 128
    net = create_network()
    tab = np.asarray(make_synthetic_series(net, n_days=400).table)
    rows = np.stack([tab[96 * (3 + b):96 * (3 + b) + 96] for b in range(B)])
    r = BatchedOPF(net).solve(rows[:, :, 71], rows[:, :, :33], rows[:, :, 33:66], rows[:, :, 66:71], np.full((B, 5), 0.0125))
    moves = [h["move"] for h in r["history"]]
    assert r["outer_iters"] <= 7 and moves[-1] < 1e-6, moves
    for a, b in zip(moves[1:], moves[2:]):
        assert b < 0.5 * a, moves                               # no plateau
    assert bool(torch.isfinite(r["objective"]).all())
