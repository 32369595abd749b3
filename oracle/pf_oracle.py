"""TEST INFRASTRUCTURE — CPU oracle for the power-flow solve.  Not product code:
only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.

PARITY UNPINNED against the reference's own solver: ``utils/pf.py`` hands a
Pyomo model to the external IPOPT binary (pf.py:101-102); pyomo, ipopt and the
xlsx network data are absent here and the reference has no tests or golden
vectors for this path.  What pins this oracle instead:
  (i)  two independent algorithms below must agree to 1e-10;
  (ii) their solution must zero the reference's own constraint expressions
       pf.py:65-94, evaluated literally in ``distflow_residuals``;
  (iii) the Baran-Wu base case lands in the literature band (min |V| ~0.913 pu
       at bus 18, losses ~202.7 kW).

Restated from the reference (file:line under /root/reference):
  pf.py:65-74   active balance   sum_in Pl - sum_out (Pl + R*Isqr) + Ps - Pnet = 0
  pf.py:76-83   reactive balance sum_in Ql - sum_out (Ql + X*Isqr) + Qs - Qnet = 0
  pf.py:85-88   Isqr[i,j] * Vsqr[j] = Pl^2 + Ql^2          (receiving-end convention)
  pf.py:90-94   Vsqr[i] - 2(R*Pl + X*Ql) - (R^2+X^2)*Isqr = Vsqr[j]
  pf.py:51-56   Vsqr = 1 and Ps,Qs free only where bus_types == 1
  pf.py:59-62   objective min sum R*Isqr: selects the low-loss (high-voltage) root
  pf.py:96-98   E_next = E_init + dt*(eta_ch*Pesc - Pesd/eta_dis), dt = 24/episode_limit
  pf.py:69-73,81-82  Pnet = Pload - Pred - Ppv + Pesc - Pesd ; Qnet = Qload - Qpv
  pf.py:108-113 returns sqrt(Vsqr), sqrt(Isqr), (Pl,Ql), E_next as dicts
"""
from __future__ import annotations

import numpy as np


# IPOPT relaxes variable bounds by bound_relax_factor * max(1, |bound|) = 1e-8 (its default; the reference sets no
# solver options, pf.py:101-102): E_next in [-1e-8, 0) still counts as inside NonNegativeReals.  Unpinned like the
# solve itself (no IPOPT here); the kernels and the C oracle use the same constant (FLEX_DOMAIN_EPS).
DOMAIN_EPS = 1e-8


class SolverFailed(Exception):
    """Stands for pf.py:104-105 ``raise Exception('Solver failed to find a solution')``."""


def _orient(net):
    """Root the radial feeder at the slack bus.  Returns (buses, idx, root, order, parent,
    key_of_child) with ``order`` a root-first BFS order."""
    buses = list(net["bus_numbers"])
    idx = {b: i for i, b in enumerate(buses)}
    slack = [b for b in buses if net["bus_types"][b] == 1]
    assert len(slack) == 1, "one slack bus expected"
    assert len(net["line_connections"]) == len(buses) - 1, "radial feeder expected"
    adj = {i: [] for i in range(len(buses))}
    for (f, t) in net["line_connections"]:
        adj[idx[f]].append((idx[t], (f, t)))
        adj[idx[t]].append((idx[f], (f, t)))
    root = idx[slack[0]]
    parent = {root: -1}
    key = {}
    order = [root]
    for u in order:
        for v, k in adj[u]:
            if v not in parent:
                parent[v] = u
                key[v] = k
                order.append(v)
    assert len(order) == len(buses), "network not connected"
    return buses, idx, root, order, parent, key


def build_ybus(net):
    """Bus admittance matrix: y = 1/(r + jx) per line, no shunts, no taps
    (create_net.py has neither)."""
    buses = list(net["bus_numbers"])
    idx = {b: i for i, b in enumerate(buses)}
    Y = np.zeros((len(buses), len(buses)), complex)
    for (f, t) in net["line_connections"]:
        y = 1.0 / (net["line_resistances"][(f, t)] + 1j * net["line_reactances"][(f, t)])
        i, j = idx[f], idx[t]
        Y[i, i] += y
        Y[j, j] += y
        Y[i, j] -= y
        Y[j, i] -= y
    return Y


def nr_polar(net, pnet, qnet, tol=1e-12, max_iter=20):
    """Algorithm (a): Newton-Raphson in polar form on the Ybus, flat start.

    ``pnet``/``qnet`` are net *loads* per bus index (injection = -load).  Unknowns
    theta[PQ], |V|[PQ]; slack held at 1∠0.  Returns (vm, va, iters).  Raises
    SolverFailed on non-convergence (mismatch inf-norm > tol after max_iter) or
    non-finite iterates — the stand-in for IPOPT status != ok (pf.py:104).
    """
    buses = list(net["bus_numbers"])
    n = len(buses)
    Y = build_ybus(net)
    slack = [i for i, b in enumerate(buses) if net["bus_types"][b] == 1][0]
    pq = np.array([i for i in range(n) if i != slack])
    psp = -np.asarray(pnet, float)
    qsp = -np.asarray(qnet, float)
    vm = np.ones(n)
    va = np.zeros(n)
    for it in range(max_iter + 1):
        V = vm * np.exp(1j * va)
        S = V * np.conj(Y @ V)
        dP = psp[pq] - S.real[pq]
        dQ = qsp[pq] - S.imag[pq]
        err = max(np.abs(dP).max(), np.abs(dQ).max())
        if not np.isfinite(err):
            raise SolverFailed("non-finite mismatch")
        if err < tol:
            return vm, va, it
        if it == max_iter:
            break
        Vd = np.diag(V)
        Id = np.diag(Y @ V)
        Vn = np.diag(V / np.abs(V))
        dS_dva = 1j * Vd @ np.conj(Id - Y @ Vd)
        dS_dvm = Vd @ np.conj(Y @ Vn) + np.conj(Id) @ Vn
        J = np.block([
            [dS_dva.real[np.ix_(pq, pq)], dS_dvm.real[np.ix_(pq, pq)]],
            [dS_dva.imag[np.ix_(pq, pq)], dS_dvm.imag[np.ix_(pq, pq)]],
        ])
        dx = np.linalg.solve(J, np.concatenate([dP, dQ]))
        va[pq] += dx[:len(pq)]
        vm[pq] += dx[len(pq):]
    raise SolverFailed(f"NR did not converge: mismatch {err:.3e}")


def distflow_sweep(net, pnet, qnet, tol=1e-13, max_iter=500):
    """Algorithm (b): backward/forward sweep on the DistFlow recursion exactly as
    pf.py writes it.  Leaf->root: P_recv[j] = Pnet[j] + sum_children (P_recv[c] +
    R_c*Isqr_c); root->leaf: Vsqr[j] = Vsqr[i] - 2(R P + X Q) - (R^2+X^2) Isqr;
    Isqr = (P^2+Q^2)/Vsqr[j].  Starting from Isqr = 0, Vsqr = 1 this iteration
    converges to the high-voltage root, the one min sum R*Isqr selects.

    Returns dict(Vsqr[n], P[n], Q[n], Isqr[n]) indexed by bus index, where entry
    j describes the line from parent(j) to j (flows *received* at j).
    """
    buses, idx, root, order, parent, key = _orient(net)
    n = len(buses)
    pnet = np.asarray(pnet, float)
    qnet = np.asarray(qnet, float)
    R = np.zeros(n)
    X = np.zeros(n)
    for v, k in key.items():
        R[v] = net["line_resistances"][k]
        X[v] = net["line_reactances"][k]
    vs = np.ones(n)
    isq = np.zeros(n)
    P = np.zeros(n)
    Q = np.zeros(n)
    for it in range(max_iter):
        P[:] = pnet
        Q[:] = qnet
        for v in reversed(order[1:]):
            u = parent[v]
            P[u] += P[v] + R[v] * isq[v]
            Q[u] += Q[v] + X[v] * isq[v]
        vs_new = np.ones(n)
        isq_new = np.zeros(n)
        for v in order[1:]:
            u = parent[v]
            # solve the pair (pf.py:85-94) for this line given Vsqr[u], P, Q:
            # Vsqr_j = Vsqr_i - 2(RP+XQ) - (R^2+X^2) * (P^2+Q^2)/Vsqr_j  (quadratic in Vsqr_j)
            a = vs_new[u] - 2.0 * (R[v] * P[v] + X[v] * Q[v])
            c = (R[v] ** 2 + X[v] ** 2) * (P[v] ** 2 + Q[v] ** 2)
            disc = a * a - 4.0 * c
            if not disc >= 0:
                raise SolverFailed("no real DistFlow root (voltage collapse)")
            vs_new[v] = 0.5 * (a + np.sqrt(disc))
            isq_new[v] = (P[v] ** 2 + Q[v] ** 2) / vs_new[v]
        delta = max(np.abs(vs_new - vs).max(), np.abs(isq_new - isq).max())
        vs, isq = vs_new, isq_new
        if not np.isfinite(delta):
            raise SolverFailed("non-finite sweep")
        if delta < tol:
            # final consistent backward pass with converged Isqr
            P[:] = pnet
            Q[:] = qnet
            for v in reversed(order[1:]):
                u = parent[v]
                P[u] += P[v] + R[v] * isq[v]
                Q[u] += Q[v] + X[v] * isq[v]
            return {"Vsqr": vs, "P": P.copy(), "Q": Q.copy(), "Isqr": isq, "iters": it + 1,
                    "parent": parent, "key": key, "root": root, "R": R, "X": X}
    raise SolverFailed(f"sweep did not converge: delta {delta:.3e}")


def distflow_residuals(net, pnet, qnet, Vsqr, Pl, Ql, Isqr):
    """Evaluate pf.py:65-94 literally.  ``Vsqr`` by bus id; ``Pl, Ql, Isqr`` by the
    reference's (from,to) line keys; ``pnet, qnet`` by bus id.  Ps, Qs are eliminated:
    they are free at the slack, fixed to 0 elsewhere (pf.py:51-56), so the balance rows
    are checked at non-slack buses only.  Returns the max abs residual."""
    L = list(net["line_connections"])
    Rr = net["line_resistances"]
    Xx = net["line_reactances"]
    worst = 0.0
    for n in net["bus_numbers"]:
        if net["bus_types"][n] == 1:
            worst = max(worst, abs(Vsqr[n] - 1.0))
            continue
        rp = (sum(Pl[(i, j)] for (i, j) in L if j == n)
              - sum(Pl[(i, j)] + Rr[(i, j)] * Isqr[(i, j)] for (i, j) in L if i == n)
              - pnet[n])
        rq = (sum(Ql[(i, j)] for (i, j) in L if j == n)
              - sum(Ql[(i, j)] + Xx[(i, j)] * Isqr[(i, j)] for (i, j) in L if i == n)
              - qnet[n])
        worst = max(worst, abs(rp), abs(rq))
    for (i, j) in L:
        worst = max(worst, abs(Isqr[(i, j)] * Vsqr[j] - (Pl[(i, j)] ** 2 + Ql[(i, j)] ** 2)))
        worst = max(worst, abs(Vsqr[i] - 2 * (Rr[(i, j)] * Pl[(i, j)] + Xx[(i, j)] * Ql[(i, j)])
                               - (Rr[(i, j)] ** 2 + Xx[(i, j)] ** 2) * Isqr[(i, j)] - Vsqr[j]))
    return worst


def branch_quantities_from_voltage(net, V):
    """From complex bus voltages (by index) derive the reference's line variables keyed by
    its (from,to) tuples: Pl,Ql = power *received* at ``to`` (pf.py:85-88 convention),
    Isqr = |I|^2."""
    buses = list(net["bus_numbers"])
    idx = {b: i for i, b in enumerate(buses)}
    Pl, Ql, Isqr = {}, {}, {}
    for (f, t) in net["line_connections"]:
        z = net["line_resistances"][(f, t)] + 1j * net["line_reactances"][(f, t)]
        cur = (V[idx[f]] - V[idx[t]]) / z
        s_recv = V[idx[t]] * np.conj(cur)
        Pl[(f, t)] = s_recv.real
        Ql[(f, t)] = s_recv.imag
        Isqr[(f, t)] = abs(cur) ** 2
    return Pl, Ql, Isqr


def solve_pf(net, pnet, qnet, tol=1e-12, max_iter=20):
    """Net loads by bus *index* -> dict with vm[n] and the reference-keyed line dicts."""
    vm, va, iters = nr_polar(net, pnet, qnet, tol, max_iter)
    V = vm * np.exp(1j * va)
    Pl, Ql, Isqr = branch_quantities_from_voltage(net, V)
    return {"vm": vm, "va": va, "iters": iters, "Pl": Pl, "Ql": Ql, "Isqr": Isqr}


def power_flow_solver_simplified(network_data, P_net, Q_net):
    """pf.py:115-192: same equations with direct net powers (dicts by bus id)."""
    buses = list(network_data["bus_numbers"])
    sol = solve_pf(network_data, [P_net[b] for b in buses], [Q_net[b] for b in buses])
    return {
        "Voltages": {b: float(sol["vm"][i]) for i, b in enumerate(buses)},
        "Currents": {k: float(np.sqrt(v)) for k, v in sol["Isqr"].items()},
        "Power Flows": {k: (float(sol["Pl"][k]), float(sol["Ql"][k])) for k in sol["Pl"]},
    }


def power_flow_solver(network_data, active_power_demand, reactive_power_demand, power_reduction,
                      pv_active_power, pv_reactive_power, ess_charging, ess_discharging,
                      initial_ess_energy, env_config=None):
    """pf.py:10-113 with the same signature and return dict.  ``env_config`` carries
    episode_limit / eta_ch / eta_dis (pf.py reads them from the YAML at import, pf.py:7-8,24,37-38)."""
    cfg = {"episode_limit": 96, "eta_ch": 0.9, "eta_dis": 0.9}
    cfg.update(env_config or {})
    B = set(network_data["buildings"])
    G = set(network_data["PVs_at_buildings"])
    K = set(network_data["ESSs_at_buildings"])
    P_net, Q_net = {}, {}
    for n in network_data["bus_numbers"]:
        P_net[n] = (active_power_demand[n]
                    - (power_reduction[n] if n in B else 0)
                    - (pv_active_power[n] if n in G else 0)
                    + (ess_charging[n] if n in K else 0)
                    - (ess_discharging[n] if n in K else 0))            # pf.py:69-73
        Q_net[n] = reactive_power_demand[n] - (pv_reactive_power[n] if n in G else 0)  # pf.py:81-82
    out = power_flow_solver_simplified(network_data, P_net, Q_net)
    dt = 24 / cfg["episode_limit"]                                      # pf.py:23-24
    out["Next ESS Energy"] = {
        k: initial_ess_energy[k] + dt * (cfg["eta_ch"] * ess_charging[k] - (1 / cfg["eta_dis"]) * ess_discharging[k])
        for k in network_data["ESSs_at_buildings"]}                     # pf.py:96-98
    # pf.py:41-45: Vsqr, Isqr, E_next are NonNegativeReals.  E_next is pinned by the equality pf.py:96-98, so a negative
    # value makes the NLP infeasible -> IPOPT status != ok -> pf.py:104-105 raises -> the env's failure path (env:314-337).
    # Vsqr = |V|^2 and Isqr = |J|^2 are non-negative by construction here; a non-finite one is a failure as well.
    if any(not (e >= -DOMAIN_EPS) for e in out["Next ESS Energy"].values()):
        raise SolverFailed("E_next outside NonNegativeReals (pf.py:45)")
    if not all(np.isfinite(v) for v in out["Voltages"].values()):
        raise SolverFailed("non-finite Vsqr (pf.py:41)")
    return out
