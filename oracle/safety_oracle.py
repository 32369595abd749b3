"""TEST INFRASTRUCTURE — CPU oracle for the safety layer.  Only tests/ and smoke() may import it.

PARITY UNPINNED against the reference's own solver: madrl/models/safemaddpg.py:176-299 builds a Pyomo QP
and hands it to Gurobi (safemaddpg.py:280-281); neither is installed and the reference holds no fixtures.
Two independent statements of the same problem pin each other instead:
  ``solve_full_qp``     the reference's 152-variable formulation (safemaddpg.py:187-277) handed to SciPy;
  ``solve_separable``   the per-building closed form of SURVEY.md App. D by active-set enumeration.
"""
from __future__ import annotations

import itertools

import numpy as np


def solve_separable(x0, pd, qd, sp, sq, beta, v_min, v_max, rho=1000.0):
    """One building: min |x-x0|^2 + rho (s_lo + s_up)  s.t.  v_min - s_lo <= c.x + d <= v_max + s_up,
    x = (pr, ch, dis, q), pr,ch,dis >= 0, s >= 0,  c = (-sp*pd, sp, -sp, sq), d = sp*pd + sq*qd + beta
    (safemaddpg.py:214-277 restricted to the variables of one building bus)."""
    x0 = np.asarray(x0, float)
    c = np.array([-sp * pd, sp, -sp, sq])
    d = sp * pd + sq * qd + beta
    best, best_obj = None, np.inf
    for side, bound, sign in (("none", 0.0, 0.0), ("up", v_max, 1.0), ("lo", v_min, -1.0)):
        for A in itertools.product([0, 1], repeat=3):
            free = np.array([1 - A[0], 1 - A[1], 1 - A[2], 1], float)
            cc = float((c * c * free).sum())
            hfree = sign * ((c * x0 * free).sum() + d - bound)
            if side == "none":
                lam = 0.0
            elif cc > 0:
                lam = min(max(2 * hfree / cc, 0.0), rho)
            else:
                lam = rho if hfree > 0 else 0.0
            x = (x0 - 0.5 * lam * sign * c) * free
            if (x[:3] < -1e-15).any():
                continue
            g = c @ x + d
            obj = ((x - x0) ** 2).sum() + rho * (max(v_min - g, 0.0) + max(g - v_max, 0.0))
            if obj < best_obj - 1e-18:
                best, best_obj = x, obj
    return best


def solve_full_qp(net, proposed, cur_pd, cur_qd, W_P, W_Q, b, v_min, v_max, rho=1000.0):
    """The reference's formulation, all 33 buses (safemaddpg.py:187-277).  ``proposed`` = dict of per-building
    arrays (pr, ch, dis, q) after parse_actions.  P_net/Q_net are eliminated through their defining equalities
    (safemaddpg.py:235-257); slacks for buses without decision variables are closed-form.  Returns the
    type-major vector [pr x n | ch x n | dis x n | q x n] of safemaddpg.py:297."""
    from scipy.optimize import minimize
    buses = list(net["bus_numbers"])
    blds = list(net["buildings"])
    n = len(blds)
    idx = [buses.index(bb) for bb in blds]
    x0 = np.concatenate([proposed["pr"], proposed["ch"], proposed["dis"], proposed["q"]])
    sP, sQ = W_P.sum(1), W_Q.sum(1)

    def vpred(x):
        out = []
        for k, bi in enumerate(idx):
            p_net = cur_pd[bi] * (1 - x[k]) + x[n + k] - x[2 * n + k]
            q_net = cur_qd[bi] + x[3 * n + k]
            out.append(sP[bi] * p_net + sQ[bi] * q_net + b[bi])
        return np.array(out)

    # variables z = [x (4n), s_lo (n), s_up (n)]; non-building buses only add a constant to the objective
    def obj(z):
        return ((z[:4 * n] - x0) ** 2).sum() + rho * z[4 * n:].sum()

    def grad(z):
        g = np.zeros_like(z)
        g[:4 * n] = 2 * (z[:4 * n] - x0)
        g[4 * n:] = rho
        return g

    cons = [{"type": "ineq", "fun": lambda z: vpred(z[:4 * n]) - v_min + z[4 * n:5 * n]},
            {"type": "ineq", "fun": lambda z: v_max + z[5 * n:] - vpred(z[:4 * n])}]
    bounds = [(0, None)] * (3 * n) + [(None, None)] * n + [(0, None)] * (2 * n)
    z0 = np.concatenate([np.maximum(x0[:3 * n], 0), x0[3 * n:], np.zeros(2 * n)])
    v0 = vpred(z0[:4 * n])
    z0[4 * n:5 * n] = np.maximum(v_min - v0, 0)
    z0[5 * n:] = np.maximum(v0 - v_max, 0)
    res = minimize(obj, z0, jac=grad, bounds=bounds, constraints=cons, method="SLSQP",
                   options={"ftol": 1e-15, "maxiter": 500})
    if not res.success:
        # SLSQP's line search can stall on the kink-free but badly scaled slack directions (rho = 1000 against
        # O(1e-3) action terms); the interior-point trust-region solver with exact derivatives is slower but robust.
        from scipy.optimize import Bounds, NonlinearConstraint
        lo = np.array([bd[0] if bd[0] is not None else -np.inf for bd in bounds])
        hi = np.array([bd[1] if bd[1] is not None else np.inf for bd in bounds])
        A = np.zeros((n, 6 * n))
        for k, bi in enumerate(idx):
            A[k, k] = -sP[bi] * cur_pd[bi]; A[k, n + k] = sP[bi]; A[k, 2 * n + k] = -sP[bi]; A[k, 3 * n + k] = sQ[bi]
        const = vpred(np.zeros(4 * n))
        from scipy.optimize import LinearConstraint
        A_lo = A.copy(); A_lo[:, 4 * n:5 * n] = np.eye(n)
        A_up = A.copy(); A_up[:, 5 * n:] = -np.eye(n)
        lc = [LinearConstraint(A_lo, v_min - const, np.inf), LinearConstraint(A_up, -np.inf, v_max - const)]
        H = np.zeros((6 * n, 6 * n)); H[:4 * n, :4 * n] = 2 * np.eye(4 * n)
        res = minimize(obj, np.clip(z0 + 1e-6, lo, hi), jac=grad, hess=lambda z: H, bounds=Bounds(lo, hi), constraints=lc,
                       method="trust-constr", options={"gtol": 1e-12, "xtol": 1e-14, "barrier_tol": 1e-12, "maxiter": 3000})
    return res.x[:4 * n], res


def kkt_violation(x, x0, pd, qd, sp, sq, beta, v_min, v_max, rho=1000.0, tol=1e-9):
    """Optimality certificate for one building's QP, independent of how ``x`` was obtained: the problem is
    convex, so x is THE minimiser iff multipliers exist with
        2(x - x0) + lam*c - mu = 0,  mu_t >= 0, mu_t*x_t = 0 (t < 3), mu_3 = 0,
        lam in [0, rho] on the upper bound (= rho beyond it), in [-rho, 0] on the lower bound, 0 strictly inside.
    Returns the largest violation of these conditions."""
    x, x0 = np.asarray(x, float), np.asarray(x0, float)
    c = np.array([-sp * pd, sp, -sp, sq])
    g = c @ x + sp * pd + sq * qd + beta
    free = [t for t in range(4) if t == 3 or x[t] > tol]
    # lam from the free coordinates (least squares over them), then check everything
    cf = c[free]
    rf = -2 * (x[free] - x0[free])
    lam = float(cf @ rf / (cf @ cf)) if (cf @ cf) > 0 else 0.0
    viol = float(np.abs(rf - lam * cf).max())                         # stationarity on free coords
    for t in range(3):
        viol = max(viol, -min(x[t], 0.0))                             # primal feasibility
        if t not in free:
            viol = max(viol, -min(2 * (x[t] - x0[t]) + lam * c[t], 0.0))   # mu_t >= 0
    if g > v_max + tol:
        viol = max(viol, abs(lam - rho))
    elif g > v_max - tol:
        viol = max(viol, -min(lam, 0.0), max(lam - rho, 0.0))
    elif g < v_min - tol:
        viol = max(viol, abs(lam + rho))
    elif g < v_min + tol:
        viol = max(viol, max(lam, 0.0), max(-rho - lam, 0.0))
    else:
        viol = max(viol, abs(lam))
    return viol
