"""TEST INFRASTRUCTURE — CPU oracle for the multi-period OPF comparator.  Not product code: only tests/ and
tools/opf_bench.py's CPU leg may import it.

PARITY UNPINNED against the reference's own solver: ``utils/opf.py`` hands a Pyomo MIQCP to Gurobi
(opf.py:152-153); pyomo and gurobipy are absent here and the reference holds no tests, fixtures or recorded
solutions for this path.  What pins this oracle instead:
  (i)   ``opf_residuals`` evaluates the reference's own constraint expressions (opf.py:96-149) and objective
        (opf.py:80-93) LITERALLY, in the reference's variables keyed the reference's way; every solution accepted
        by a test must zero them;
  (ii)  ``solve_reduced`` finds a KKT point of that program with SciPy SLSQP in the space of the controls (the
        network state follows from them through the power-flow oracle, which itself zeroes pf.py:65-94);
  (iii) ``first_order_gap`` bounds how much any feasible first-order move could still gain at a candidate.

Restated from the reference (file:line under /root/reference):
  opf.py:27-28   delta_t = 24 / episode_limit
  opf.py:46      Pred_max[b,t] = Pload[b,t] * max_power_reduction
  opf.py:54-65   bounds: 0<=Pred<=Pred_max, Qpv free, 0<=Pesc<=Pch_max, 0<=Pesd<=Pdis_max, Emin<=E<=Emax,
                 Isqr>=0, Vsqr>=0, charging_indicator binary
  opf.py:68-75   Vsqr = 1 at the substation; Ps = Qs = 0 elsewhere
  opf.py:79-93   maximise sum_t dt*( lambda_t*sum Pred - pv_cost*sum Qpv - ess_cost*sum(Pesc+Pesd)
                                     - sum R*Isqr - discomfort*sum Pred^2 )
  opf.py:96-107  active balance:  sum_in Pl - sum_out (Pl + R Isqr) + Pesd - Pesc + Ppv + Ps - Pload + Pred = 0
  opf.py:109-116 reactive balance: sum_in Ql - sum_out (Ql + X Isqr) + Qpv + Qs - Qload = 0
  opf.py:118-120 |Qpv| <= tan(acos(cos_phi_max)) * Ppv
  opf.py:122-125 Vsqr_i - 2(R Pl + X Ql) - (R^2+X^2) Isqr = Vsqr_j
  opf.py:127-129 Isqr * Vsqr_j = Pl^2 + Ql^2
  opf.py:131-133 Isqr <= Imax^2
  opf.py:135-137 Vmin^2 <= Vsqr <= Vmax^2
  opf.py:139-148 E[k,1] = E_init;  E[k,t] = E[k,t-1] + dt*(eta_ch*Pesc[k,t] - Pesd[k,t]/eta_dis)   (t >= 2: the
                 first period's charging never reaches the energy balance)
  opf.py:150-156 Pesc <= Pch_max*z, Pesd <= Pdis_max*(1-z), z binary
The binaries are relaxed here: with ess_cost > 0 charging and discharging at once only burns energy and money, so the
relaxation is tight unless an over-voltage wants extra load; ``opf_residuals`` reports min(Pesc, Pesd) so that a test
can insist on complementarity.
"""
from __future__ import annotations

from math import acos, tan

import numpy as np

from . import pf_oracle

DEFAULT_CFG = dict(episode_limit=96, v_min=0.9, v_max=1.1, pv_cost=0.05, ess_cost=0.03, discomfort_coeff=0.15,
                   eta_ch=0.9, eta_dis=0.9, max_power_reduction=0.5, e_min=0.0, e_max=0.025, p_ch_max=0.005,
                   p_dis_max=0.005, cos_phi_max=0.95)


def _cfg(cfg):
    out = dict(DEFAULT_CFG)
    out.update(cfg or {})
    return out


def state_from_controls(net, cfg, pd, qd, ppv, e0, pred, qpv, ch, dis):
    """Controls [T, na] -> the reference's remaining variables: per period the DistFlow state of the net loads
    (opf.py:96-116 define them: Pnet = Pload - Pred - Ppv + Pesc - Pesd, Qnet = Qload - Qpv) and the energy chain."""
    cfg = _cfg(cfg)
    buses = list(net["bus_numbers"])
    idx = {b: i for i, b in enumerate(buses)}
    bld = [idx[b] for b in net["buildings"]]
    T = pd.shape[0]
    dt = 24.0 / cfg["episode_limit"]
    vsqr = np.zeros((T, len(buses)))
    lines = list(net["line_connections"])
    pl = np.zeros((T, len(lines)))
    ql = np.zeros((T, len(lines)))
    isqr = np.zeros((T, len(lines)))
    for t in range(T):
        pnet = pd[t].copy()
        qnet = qd[t].copy()
        pnet[bld] += -pred[t] - ppv[t] + ch[t] - dis[t]
        qnet[bld] += -qpv[t]
        s = pf_oracle.solve_pf(net, pnet, qnet)
        vsqr[t] = s["vm"] ** 2
        for k, key in enumerate(lines):
            pl[t, k], ql[t, k], isqr[t, k] = s["Pl"][key], s["Ql"][key], s["Isqr"][key]
    e = np.zeros_like(ch)
    e[0] = e0
    for t in range(1, T):
        e[t] = e[t - 1] + dt * (cfg["eta_ch"] * ch[t] - dis[t] / cfg["eta_dis"])
    return dict(Vsqr=vsqr, Pl=pl, Ql=ql, Isqr=isqr, E=e)


def opf_objective(net, cfg, price, sol):
    """opf.py:80-93, literally (losses over the reference's line keys)."""
    cfg = _cfg(cfg)
    T = len(price)
    dt = 24.0 / cfg["episode_limit"]
    lines = list(net["line_connections"])
    total = 0.0
    for t in range(T):
        total += dt * (sum(price[t] * sol["Pred"][t][b] for b in range(sol["Pred"].shape[1]))
                       - sum(cfg["pv_cost"] * sol["Qpv"][t][g] for g in range(sol["Qpv"].shape[1]))
                       - sum(cfg["ess_cost"] * (sol["Pesc"][t][k] + sol["Pesd"][t][k]) for k in range(sol["Pesc"].shape[1]))
                       - sum(net["line_resistances"][key] * sol["Isqr"][t][i] for i, key in enumerate(lines))
                       - sum(cfg["discomfort_coeff"] * sol["Pred"][t][b] ** 2 for b in range(sol["Pred"].shape[1])))
    return total


def opf_residuals(net, cfg, pd, qd, ppv, e0, sol):
    """Max violation per constraint family of opf.py:96-156 (and the bounds opf.py:54-65), evaluated the reference's
    way: buses by id, lines by (from, to).  ``sol``: Pred, Qpv, Pesc, Pesd, E [T, na]; Vsqr [T, n_bus]; Pl, Ql, Isqr
    [T, n_line] in the order of net['line_connections']."""
    cfg = _cfg(cfg)
    buses = list(net["bus_numbers"])
    idx = {b: i for i, b in enumerate(buses)}
    L = list(net["line_connections"])
    lk = {key: i for i, key in enumerate(L)}
    R, X, Imax = net["line_resistances"], net["line_reactances"], net["max_line_currents"]
    B, G, K = list(net["buildings"]), list(net["PVs_at_buildings"]), list(net["ESSs_at_buildings"])
    T = pd.shape[0]
    dt = 24.0 / cfg["episode_limit"]
    tanphi = tan(acos(cfg["cos_phi_max"]))
    out = dict(active=0.0, reactive=0.0, vdrop=0.0, current_def=0.0, current_lim=0.0, v_lim=0.0, qpv_lim=0.0,
               energy=0.0, e_lim=0.0, box=0.0, simultaneous=0.0, slack_v=0.0)
    for t in range(T):
        for n in buses:
            if net["bus_types"][n] == 1:
                out["slack_v"] = max(out["slack_v"], abs(sol["Vsqr"][t][idx[n]] - 1.0))
                continue                                           # Ps, Qs free there (opf.py:68-75)
            ra = (sum(sol["Pl"][t][lk[(i, j)]] for (i, j) in L if j == n)
                  - sum(sol["Pl"][t][lk[(i, j)]] + R[(i, j)] * sol["Isqr"][t][lk[(i, j)]] for (i, j) in L if i == n)
                  + sum(sol["Pesd"][t][k] for k, bus in enumerate(K) if bus == n)
                  - sum(sol["Pesc"][t][k] for k, bus in enumerate(K) if bus == n)
                  + sum(ppv[t][g] for g, bus in enumerate(G) if bus == n)
                  - pd[t][idx[n]]
                  + sum(sol["Pred"][t][b] for b, bus in enumerate(B) if bus == n))
            rr = (sum(sol["Ql"][t][lk[(i, j)]] for (i, j) in L if j == n)
                  - sum(sol["Ql"][t][lk[(i, j)]] + X[(i, j)] * sol["Isqr"][t][lk[(i, j)]] for (i, j) in L if i == n)
                  + sum(sol["Qpv"][t][g] for g, bus in enumerate(G) if bus == n)
                  - qd[t][idx[n]])
            out["active"] = max(out["active"], abs(ra))
            out["reactive"] = max(out["reactive"], abs(rr))
            v = sol["Vsqr"][t][idx[n]]
            out["v_lim"] = max(out["v_lim"], cfg["v_min"] ** 2 - v, v - cfg["v_max"] ** 2)
        for (i, j) in L:
            k = lk[(i, j)]
            out["vdrop"] = max(out["vdrop"], abs(sol["Vsqr"][t][idx[i]] - 2 * (R[(i, j)] * sol["Pl"][t][k] + X[(i, j)] * sol["Ql"][t][k])
                                                 - (R[(i, j)] ** 2 + X[(i, j)] ** 2) * sol["Isqr"][t][k] - sol["Vsqr"][t][idx[j]]))
            out["current_def"] = max(out["current_def"], abs(sol["Isqr"][t][k] * sol["Vsqr"][t][idx[j]]
                                                             - (sol["Pl"][t][k] ** 2 + sol["Ql"][t][k] ** 2)))
            out["current_lim"] = max(out["current_lim"], sol["Isqr"][t][k] - Imax[(i, j)] ** 2)
        for g, bus in enumerate(G):
            out["qpv_lim"] = max(out["qpv_lim"], abs(sol["Qpv"][t][g]) - tanphi * ppv[t][g])
        for k in range(len(K)):
            if t == 0:
                res = sol["E"][t][k] - e0[k]
            else:
                res = sol["E"][t][k] - (sol["E"][t - 1][k] + dt * (cfg["eta_ch"] * sol["Pesc"][t][k]
                                                                    - (1 / cfg["eta_dis"]) * sol["Pesd"][t][k]))
            out["energy"] = max(out["energy"], abs(res))
            out["e_lim"] = max(out["e_lim"], cfg["e_min"] - sol["E"][t][k], sol["E"][t][k] - cfg["e_max"])
            out["simultaneous"] = max(out["simultaneous"], min(sol["Pesc"][t][k], sol["Pesd"][t][k]))
            out["box"] = max(out["box"], -sol["Pesc"][t][k], sol["Pesc"][t][k] - cfg["p_ch_max"],
                             -sol["Pesd"][t][k], sol["Pesd"][t][k] - cfg["p_dis_max"])
        for b, bus in enumerate(B):
            out["box"] = max(out["box"], -sol["Pred"][t][b],
                             sol["Pred"][t][b] - pd[t][idx[bus]] * cfg["max_power_reduction"])
    return out


# ---- reduced-space program: controls x[T, 4, na] = (Pred, Qpv, Pesc, Pesd) -------------------------------------
class ReducedOPF:
    def __init__(self, net, cfg, price, pd, qd, ppv, e0):
        self.net, self.cfg = net, _cfg(cfg)
        self.price, self.pd, self.qd, self.ppv, self.e0 = (np.asarray(a, float) for a in (price, pd, qd, ppv, e0))
        self.T, self.na = self.pd.shape[0], self.ppv.shape[1]
        buses = list(net["bus_numbers"])
        self.bld = [buses.index(b) for b in net["buildings"]]
        self.nonslack = [i for i, b in enumerate(buses) if net["bus_types"][b] != 1]
        self.imax2 = np.array([net["max_line_currents"][k] ** 2 for k in net["line_connections"]])
        self.rline = np.array([net["line_resistances"][k] for k in net["line_connections"]])
        self.dt = 24.0 / self.cfg["episode_limit"]
        self.tanphi = tan(acos(self.cfg["cos_phi_max"]))

    def split(self, x):
        x = np.asarray(x, float).reshape(self.T, 4, self.na)
        return x[:, 0], x[:, 1], x[:, 2], x[:, 3]

    def bounds(self):
        c = self.cfg
        lo = np.zeros((self.T, 4, self.na))
        hi = np.zeros((self.T, 4, self.na))
        hi[:, 0] = self.pd[:, self.bld] * c["max_power_reduction"]
        lo[:, 1], hi[:, 1] = -self.tanphi * self.ppv, self.tanphi * self.ppv
        hi[:, 2], hi[:, 3] = c["p_ch_max"], c["p_dis_max"]
        return lo.ravel(), hi.ravel()

    def period_state(self, t, xt):
        """One period's network response: (Vsqr non-slack, Isqr, losses)."""
        pnet = self.pd[t].copy()
        qnet = self.qd[t].copy()
        pnet[self.bld] += -xt[0] - self.ppv[t] + xt[2] - xt[3]
        qnet[self.bld] += -xt[1]
        s = pf_oracle.solve_pf(self.net, pnet, qnet)
        isqr = np.array([s["Isqr"][k] for k in self.net["line_connections"]])
        return s["vm"][self.nonslack] ** 2, isqr, float(self.rline @ isqr)

    def energy(self, x):
        _, _, ch, dis = self.split(x)
        inc = self.dt * (self.cfg["eta_ch"] * ch - dis / self.cfg["eta_dis"])
        inc[0] = 0.0                                              # opf.py:140-142
        return self.e0[None, :] + np.cumsum(inc, axis=0)

    def evaluate(self, x, with_jac=False, h=1e-6):
        """objective (to MAXIMISE), network constraint values g(x) (Vsqr, Isqr stacked per period) and, on request,
        their derivatives by central differences — a control of period t only moves period t."""
        c = self.cfg
        pred, qpv, ch, dis = self.split(x)
        xs = np.asarray(x, float).reshape(self.T, 4, self.na)
        f = 0.0
        g = []
        df = np.zeros_like(xs)
        jac = []
        for t in range(self.T):
            v2, i2, loss = self.period_state(t, xs[t])
            f += self.dt * (self.price[t] * pred[t].sum() - c["pv_cost"] * qpv[t].sum() - c["ess_cost"] * (ch[t] + dis[t]).sum()
                            - loss - c["discomfort_coeff"] * (pred[t] ** 2).sum())
            g.append(np.concatenate([v2, i2]))
            if with_jac:
                jt = np.zeros((len(g[-1]), 4, self.na))
                dloss = np.zeros((4, self.na))
                for a in range(4):
                    for k in range(self.na):
                        xp, xm = xs[t].copy(), xs[t].copy()
                        xp[a, k] += h
                        xm[a, k] -= h
                        vp, ip, lp = self.period_state(t, xp)
                        vm, im, lm = self.period_state(t, xm)
                        jt[:, a, k] = (np.concatenate([vp, ip]) - np.concatenate([vm, im])) / (2 * h)
                        dloss[a, k] = (lp - lm) / (2 * h)
                jac.append(jt)
                df[t, 0] = self.dt * (self.price[t] - 2 * c["discomfort_coeff"] * pred[t])
                df[t, 1] = -self.dt * c["pv_cost"]
                df[t, 2] = -self.dt * c["ess_cost"]
                df[t, 3] = -self.dt * c["ess_cost"]
                df[t] -= self.dt * dloss
        return f, g, (df, jac) if with_jac else None


def solve_reduced(net, cfg, price, pd, qd, ppv, e0, x0=None, maxiter=200, ftol=1e-13):
    """KKT point of the relaxed program by SLSQP over the controls.  Returns (x[T,4,na], objective, info)."""
    from scipy.optimize import minimize

    P = ReducedOPF(net, cfg, price, pd, qd, ppv, e0)
    c = P.cfg
    lo, hi = P.bounds()
    T, na = P.T, P.na
    if x0 is None:
        x0 = np.zeros((T, 4, na))
        x0[:, 0] = np.clip(np.asarray(price, float)[:, None] / (2 * c["discomfort_coeff"]), 0, hi.reshape(T, 4, na)[:, 0])
    x0 = np.clip(np.asarray(x0, float).ravel(), lo, hi)
    n_v = len(P.nonslack)
    g_lo = np.concatenate([np.full(n_v, c["v_min"] ** 2), np.full(len(P.imax2), -np.inf)])
    g_hi = np.concatenate([np.full(n_v, c["v_max"] ** 2), P.imax2])
    cache = {}

    def ev(x):
        key = x.tobytes()
        if key not in cache:
            cache.clear()
            cache[key] = P.evaluate(x, with_jac=True)
        return cache[key]

    def fun(x):
        return -ev(x)[0]

    def jac(x):
        return -ev(x)[2][0].ravel()

    def cons(x):                                   # >= 0
        g = ev(x)[1]
        e = P.energy(x)
        parts = []
        for t in range(T):
            parts += [g[t][:n_v] - g_lo[:n_v], g_hi - g[t]]
        parts += [(e - c["e_min"]).ravel(), (c["e_max"] - e).ravel()]
        return np.concatenate(parts)

    def cons_jac(x):
        _, _, (_, jt) = ev(x)
        rows = []
        for t in range(T):
            full = np.zeros((jt[t].shape[0], T, 4, na))
            full[:, t] = jt[t]
            full = full.reshape(jt[t].shape[0], -1)
            rows += [full[:n_v], -full]
        je = np.zeros((T, na, T, 4, na))
        for t in range(1, T):
            for s in range(1, t + 1):
                for k in range(na):
                    je[t, k, s, 2, k] = P.dt * c["eta_ch"]
                    je[t, k, s, 3, k] = -P.dt / c["eta_dis"]
        je = je.reshape(T * na, -1)
        rows += [je, -je]
        return np.vstack(rows)

    res = minimize(fun, x0, jac=jac, bounds=list(zip(lo, hi)), method="SLSQP",
                   constraints=[dict(type="ineq", fun=cons, jac=cons_jac)], options=dict(maxiter=maxiter, ftol=ftol))
    x = res.x.reshape(T, 4, na)
    return x, -res.fun, dict(success=bool(res.success), message=str(res.message), nit=int(res.nit), problem=P)


def solution_dict(net, cfg, pd, qd, ppv, e0, x):
    """Controls -> every variable opf.py:160-189 reports (arrays; ``opf_residuals`` takes this)."""
    x = np.asarray(x, float)
    st = state_from_controls(net, cfg, np.asarray(pd, float), np.asarray(qd, float), np.asarray(ppv, float),
                             np.asarray(e0, float), x[:, 0], x[:, 1], x[:, 2], x[:, 3])
    return dict(Pred=x[:, 0], Qpv=x[:, 1], Pesc=x[:, 2], Pesd=x[:, 3], **st)


def first_order_gap(net, cfg, price, pd, qd, ppv, e0, x, active_tol=1e-7):
    """How much objective a feasible first-order move could still gain at ``x``: the optimum of the LINEARISED program
    over the trust box |dx| <= 1 % of each control's range (scipy linprog / HiGHS).  ~0 at a KKT point."""
    from scipy.optimize import linprog

    P = ReducedOPF(net, cfg, price, pd, qd, ppv, e0)
    c = P.cfg
    T, na = P.T, P.na
    x = np.asarray(x, float).reshape(T, 4, na)
    f, g, (df, jt) = P.evaluate(x.ravel(), with_jac=True)
    lo, hi = P.bounds()
    rng = 0.01 * np.maximum(hi - lo, 1e-6)
    lb = np.maximum(lo - x.ravel(), -rng)
    ub = np.minimum(hi - x.ravel(), rng)
    n_v = len(P.nonslack)
    A, b = [], []
    for t in range(T):
        full = np.zeros((jt[t].shape[0], T, 4, na))
        full[:, t] = jt[t]
        full = full.reshape(jt[t].shape[0], -1)
        A += [full[:n_v], -full[:n_v], full[n_v:]]
        b += [c["v_max"] ** 2 - g[t][:n_v], g[t][:n_v] - c["v_min"] ** 2, P.imax2 - g[t][n_v:]]
    e = P.energy(x.ravel())
    je = np.zeros((T, na, T, 4, na))
    for t in range(1, T):
        for s in range(1, t + 1):
            for k in range(na):
                je[t, k, s, 2, k] = P.dt * c["eta_ch"]
                je[t, k, s, 3, k] = -P.dt / c["eta_dis"]
    je = je.reshape(T * na, -1)
    A += [je, -je]
    b += [(c["e_max"] - e).ravel(), (e - c["e_min"]).ravel()]
    A = np.vstack(A)
    b = np.maximum(np.concatenate(b), 0.0)             # a candidate that violates by round-off still admits dx = 0
    res = linprog(-df.ravel(), A_ub=A, b_ub=b, bounds=list(zip(lb, ub)), method="highs")
    return float(-res.fun) if res.status == 0 else float("nan")
