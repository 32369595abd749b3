"""TEST INFRASTRUCTURE ONLY (never imported by the product path).

The reference's power flow AS AN NLP: the variables, bounds, objective and constraints of utils/pf.py:39-94 written out
verbatim — Vsqr[N] >= 0 (slack fixed to 1, pf.py:51-56), Pl[L], Ql[L] free, Isqr[L] >= 0, Ps / Qs free at the slack and
fixed to 0 elsewhere; minimise sum R * Isqr (pf.py:59-62) subject to the active / reactive balances (pf.py:65-83), the
current definition Isqr * Vsqr[j] = Pl^2 + Ql^2 (pf.py:85-88) and the voltage drop (pf.py:90-94) — and handed to SciPy's
general NLP solvers instead of IPOPT (which is not in this image).  Pyomo declares these variables without initial
values, so IPOPT starts from 0 pushed 1e-2 inside the bounds (`bound_push`); `ipopt_default_start` is that point.

What it pins (tests/test_oracle_cpu.py): the NLP formulation of the reference, solved by two algorithms of a different
family than ours (SLSQP: sequential quadratic programming; trust-constr: trust-region interior point) from the
reference's own start, lands on the voltages of oracle/pf_oracle.py — the fixed point the HIP kernels converge to — and
not on one of the low-voltage roots of the same equations.  It is still not a run of the reference: parity for the
power flow stays "unpinned" in the sense of DESIGN.md §5."""
import numpy as np
from scipy.optimize import minimize


class ReferenceNLP:
    def __init__(self, net, pnet, qnet):
        buses, lines = net["bus_numbers"], net["line_connections"]
        bi = {b: i for i, b in enumerate(buses)}
        self.nb, self.nl = len(buses), len(lines)
        self.R = np.array([net["line_resistances"][l] for l in lines])
        self.X = np.array([net["line_reactances"][l] for l in lines])
        self.fr = np.array([bi[l[0]] for l in lines])
        self.to = np.array([bi[l[1]] for l in lines])
        self.slack = [i for i, b in enumerate(buses) if net["bus_types"][b] == 1][0]
        self.free_v = [i for i in range(self.nb) if i != self.slack]
        self.P, self.Q = np.asarray(pnet, float), np.asarray(qnet, float)      # net load per bus (pf.py:69-73, 81-82)
        nv, nl = len(self.free_v), self.nl
        self.n = nv + 3 * nl + 2
        lb = np.full(self.n, -np.inf)
        lb[:nv] = 0.0                                   # Vsqr within NonNegativeReals (pf.py:41)
        lb[nv + 2 * nl:nv + 3 * nl] = 0.0                # Isqr within NonNegativeReals (pf.py:44)
        self.lb = lb

    def unpack(self, x):
        nv, nl = len(self.free_v), self.nl
        V = np.ones(self.nb)
        V[self.free_v] = x[:nv]
        return V, x[nv:nv + nl], x[nv + nl:nv + 2 * nl], x[nv + 2 * nl:nv + 3 * nl], x[-2], x[-1]

    def objective(self, x):
        return float((self.R * self.unpack(x)[3]).sum())                       # pf.py:59-62

    def constraints(self, x):
        V, Pl, Ql, I, Ps, Qs = self.unpack(x)
        pb, qb = np.zeros(self.nb), np.zeros(self.nb)
        np.add.at(pb, self.to, Pl)                                             # pf.py:67
        np.add.at(pb, self.fr, -(Pl + self.R * I))                             # pf.py:68
        np.add.at(qb, self.to, Ql)                                             # pf.py:78
        np.add.at(qb, self.fr, -(Ql + self.X * I))                             # pf.py:79
        pb -= self.P
        qb -= self.Q
        pb[self.slack] += Ps                                                   # Ps, Qs are free at the slack only (pf.py:51-56)
        qb[self.slack] += Qs
        cur = I * V[self.to] - (Pl ** 2 + Ql ** 2)                             # pf.py:85-88
        vd = V[self.fr] - 2 * (self.R * Pl + self.X * Ql) - (self.R ** 2 + self.X ** 2) * I - V[self.to]   # pf.py:90-94
        return np.concatenate([pb, qb, cur, vd])

    def ipopt_default_start(self):
        return np.where(np.isfinite(self.lb), 1e-2, 0.0)

    def solve(self, method="SLSQP", x0=None):
        x0 = self.ipopt_default_start() if x0 is None else x0
        opts = dict(maxiter=2000, ftol=1e-14) if method == "SLSQP" else dict(maxiter=5000, gtol=1e-12, xtol=1e-15)
        r = minimize(self.objective, x0, method=method, bounds=list(zip(self.lb, [None] * self.n)),
                     constraints=[{"type": "eq", "fun": self.constraints}], options=opts)
        V, Pl, Ql, I, _, _ = self.unpack(r.x)
        return dict(success=bool(r.success), vm=np.sqrt(np.maximum(V, 0.0)), Pl=Pl, Ql=Ql, Isqr=I, objective=float(r.fun),
                    residual=float(np.abs(self.constraints(r.x)).max()))
