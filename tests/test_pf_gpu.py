"""GPU parity: batched HIP power flow (through the C ABI) vs the CPU oracle.
Tolerance: |dV| <= 1e-10 pu (north_star asks 1e-6; the fixed point is the same to rounding)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL_V = 1e-10


SOLVERS = [0, 2]   # FLEX_SOLVER_TREE (Newton + tree elimination), FLEX_SOLVER_SWEEP (sweeps, Newton-verified)


def _solve(net, p, q, **kw):
    import torch
    from safe_marl_amd.flex_env import pf_solve_batch
    out = pf_solve_batch(net, torch.from_numpy(p).cuda(), torch.from_numpy(q).cuda(), **kw)
    torch.cuda.synchronize()
    return {k: v.cpu().numpy() for k, v in out.items()}


def test_base_case_literature_band(net, base_loads):
    """Baran-Wu base case: min |V| ~ 0.9131 pu at bus 18, losses ~ 202.7 kW."""
    p, q = base_loads
    out = _solve(net, p[None], q[None], want_branch=True)
    v = out["v"][0]
    assert not out["failed"][0]
    assert abs(v.min() - 0.91309) < 1e-4 and net["bus_numbers"][int(v.argmin())] == 18
    from safe_marl_amd.network import build_tables
    t = build_tables(net)
    loss_kw = float((t.r * out["isqr"][0]).sum() * 1000)
    assert abs(loss_kw - 202.677) < 0.05


@pytest.mark.parametrize("solver", SOLVERS)
def test_random_injections_match_oracle(net, base_loads, solver):
    from oracle import pf_oracle
    p, q = base_loads
    rng = np.random.default_rng(7)
    n = 512
    P = p[None] * rng.uniform(0.0, 1.6, (n, len(p)))
    Q = q[None] * rng.uniform(-0.5, 1.6, (n, len(p)))
    # reverse flows: PV-heavy cases
    P[::5] -= rng.uniform(0, 0.3, (len(P[::5]), len(p))) * (np.arange(len(p)) > 0)
    out = _solve(net, P, Q, want_branch=True, solver=solver)
    assert out["failed"].sum() == 0
    if solver == 2:   # sweeps did the work; Newton only verified (iters = newton + 1000*sweeps)
        assert (out["iters"] % 1000 <= 1).all() and (out["iters"] // 1000 >= 3).all()
    worst = 0.0
    for i in range(0, n, 4):
        sol = pf_oracle.solve_pf(net, P[i], Q[i])
        worst = max(worst, np.abs(sol["vm"] - out["v"][i]).max())
        # line quantities in the reference's (from,to) keys, receiving-end convention
        from safe_marl_amd.network import build_tables
        t = build_tables(net)
        for b in range(t.n_bus):
            key = t.line_of_bus[b]
            if key is None:
                continue
            assert abs(sol["Isqr"][key] - out["isqr"][i, b]) < 1e-9
            assert abs(sol["Pl"][key] - out["pl"][i, b]) < 1e-9
            assert abs(sol["Ql"][key] - out["ql"][i, b]) < 1e-9
    assert worst < TOL_V


def test_distflow_residuals_of_gpu_solution(net, base_loads):
    """The GPU solution must zero the reference's own constraints pf.py:65-94."""
    from oracle import pf_oracle
    from safe_marl_amd.network import build_tables
    p, q = base_loads
    rng = np.random.default_rng(11)
    P = p[None] * rng.uniform(0.2, 1.4, (16, len(p)))
    Q = q[None] * rng.uniform(0.2, 1.4, (16, len(p)))
    out = _solve(net, P, Q, want_branch=True)
    t = build_tables(net)
    buses = net["bus_numbers"]
    for i in range(16):
        Vs = {b: out["v"][i, k] ** 2 for k, b in enumerate(buses)}
        Pl = {t.line_of_bus[k]: out["pl"][i, k] for k in range(t.n_bus) if t.line_of_bus[k]}
        Ql = {t.line_of_bus[k]: out["ql"][i, k] for k in range(t.n_bus) if t.line_of_bus[k]}
        Is = {t.line_of_bus[k]: out["isqr"][i, k] for k in range(t.n_bus) if t.line_of_bus[k]}
        res = pf_oracle.distflow_residuals(net, dict(zip(buses, P[i])), dict(zip(buses, Q[i])), Vs, Pl, Ql, Is)
        assert res < 1e-11


@pytest.mark.parametrize("solver", SOLVERS)
def test_voltage_collapse_reports_failed(net, base_loads, solver):
    """No power-flow solution exists at 20x load: failure is data, not an exception (env:314-337)."""
    p, q = base_loads
    out = _solve(net, np.stack([p, 20 * p]), np.stack([q, 20 * q]), solver=solver)
    assert list(out["failed"]) == [0, 1]


def test_heavy_load_near_collapse_both_solvers_agree(net, base_loads):
    """3.2x load (min |V| ~ 0.6 pu): sweeps converge slowly or stall, the Newton fallback must finish."""
    from oracle import pf_oracle
    p, q = base_loads
    ref = pf_oracle.nr_polar(net, 3.2 * p, 3.2 * q, max_iter=30)[0]
    for solver in SOLVERS:
        out = _solve(net, 3.2 * p[None], 3.2 * q[None], solver=solver)
        assert not out["failed"][0]
        assert np.abs(out["v"][0] - ref).max() < 1e-9


def test_other_topology_star_and_chain():
    """Trees other than IEEE-33: a 3-way branch at the root's child and a bus-number order that is not a DFS order."""
    from oracle import pf_oracle
    from safe_marl_amd.network import create_network
    nodes = [(b, 1 if b == 1 else 0, 50.0 + 10 * b, 20.0 + 3 * b) for b in range(1, 10)]
    nodes[0] = (1, 1, 0.0, 0.0)
    lines = [(1, 5, 0.3, 0.2, 400), (5, 2, 0.5, 0.4, 400), (5, 9, 0.4, 0.3, 400), (5, 3, 0.6, 0.2, 400),
             (3, 7, 0.7, 0.5, 400), (9, 4, 0.2, 0.2, 400), (4, 6, 0.9, 0.8, 400), (2, 8, 0.3, 0.1, 400)]
    net9 = create_network({"buildings": [2, 4], "pv_nodes": [2, 4], "ess_nodes": [2, 4]}, nodes, lines)
    buses = net9["bus_numbers"]
    p = np.array([net9["active_power_demand"][b] for b in buses])
    q = np.array([net9["reactive_power_demand"][b] for b in buses])
    sol = pf_oracle.solve_pf(net9, p, q)
    for solver in SOLVERS:
        out = _solve(net9, p[None], q[None], solver=solver)
        assert np.abs(out["v"][0] - sol["vm"]).max() < TOL_V


def test_dense_lu_newton_variant_matches_oracle(net, base_loads):
    """FLEX_SOLVER_DENSE: Newton with a dense LU of the 64 x 64 Jacobian in LDS (the north-star's reference variant)."""
    from oracle import pf_oracle
    p, q = base_loads
    rng = np.random.default_rng(5)
    n = 96
    P = p[None] * rng.uniform(0.0, 1.6, (n, len(p)))
    Q = q[None] * rng.uniform(-0.5, 1.6, (n, len(p)))
    out = _solve(net, P, Q, solver=1)
    assert out["failed"].sum() == 0 and out["iters"].max() <= 6
    for i in range(0, n, 8):
        assert np.abs(pf_oracle.nr_polar(net, P[i], Q[i])[0] - out["v"][i]).max() < TOL_V
    tree = _solve(net, P, Q, solver=0)
    assert np.array_equal(out["iters"], tree["iters"] % 1000)          # same Newton iteration, different linear solver
    assert np.abs(out["v"] - tree["v"]).max() < 1e-13


def _random_feeder(n_bus, seed, buildings):
    """A random radial feeder with unordered bus numbering (so lane order != bus order) and branching up to 4."""
    from safe_marl_amd.network import create_network
    rng = np.random.default_rng(seed)
    ids = list(range(1, n_bus + 1))
    order = [1] + list(rng.permutation(ids[1:]))
    lines, nchild = [], {b: 0 for b in ids}
    for k in range(1, n_bus):
        cands = [b for b in order[:k] if nchild[b] < (4 if b != 1 else 2)]
        par = int(cands[-1] if rng.random() < 0.6 else rng.choice(cands))     # mostly chains, sometimes branches
        nchild[par] += 1
        lines.append((par, int(order[k]), float(rng.uniform(0.1, 0.8)), float(rng.uniform(0.05, 0.6)), 400.0))
    nodes = [(b, 1 if b == 1 else 0, 0.0 if b == 1 else float(rng.uniform(20, 120)), 0.0 if b == 1 else float(rng.uniform(5, 60)))
             for b in ids]
    return create_network({"buildings": buildings, "pv_nodes": buildings, "ess_nodes": buildings}, nodes, lines)


@pytest.mark.parametrize("n_bus,seed", [(45, 1), (64, 2), (20, 3)])
def test_other_feeders_one_env_per_wavefront_path(n_bus, seed):
    """Feeders with more than 32 PQ buses run one environment per wavefront (EPW = 1); smaller ones two."""
    from oracle import pf_oracle
    netx = _random_feeder(n_bus, seed, [2, 3])
    buses = netx["bus_numbers"]
    p = np.array([netx["active_power_demand"][b] for b in buses])
    q = np.array([netx["reactive_power_demand"][b] for b in buses])
    rng = np.random.default_rng(seed)
    P = p[None] * rng.uniform(0.2, 1.3, (9, n_bus))
    Q = q[None] * rng.uniform(0.2, 1.3, (9, n_bus))
    for solver in SOLVERS:
        out = _solve(netx, P, Q, solver=solver, want_branch=True)
        assert out["failed"].sum() == 0
        for i in range(9):
            sol = pf_oracle.solve_pf(netx, P[i], Q[i])
            assert np.abs(out["v"][i] - sol["vm"]).max() < TOL_V


def test_full_size_batch_satisfies_reference_constraints(net, base_loads):
    """4096 solves at once: the reference's own DistFlow constraints (pf.py:65-94) evaluated in vectorised NumPy on the
    GPU outputs — a size-independent residual — and linearity of the loss-free part (sum of injections = slack power - losses)."""
    from safe_marl_amd.network import build_tables
    p, q = base_loads
    rng = np.random.default_rng(9)
    n = 4096
    P = p[None] * rng.uniform(0.0, 1.5, (n, 33))
    Q = q[None] * rng.uniform(-0.3, 1.5, (n, 33))
    out = _solve(net, P, Q, want_branch=True)
    assert out["failed"].sum() == 0
    t = build_tables(net)
    vs = out["v"] ** 2
    par = t.parent
    ch = [np.where(par == b)[0] for b in range(t.n_bus)]
    R, X = t.r[None], t.x[None]
    pl, ql, isq = out["pl"], out["ql"], out["isqr"]
    worst = 0.0
    for b in range(t.n_bus):
        if b == t.slack:
            continue
        # pf.py:65-83 at bus b: inflow - sum_out (flow + loss) - net load = 0
        rp = pl[:, b] - sum(pl[:, c] + t.r[c] * isq[:, c] for c in ch[b]) - P[:, b]
        rq = ql[:, b] - sum(ql[:, c] + t.x[c] * isq[:, c] for c in ch[b]) - Q[:, b]
        worst = max(worst, np.abs(rp).max(), np.abs(rq).max())
        # pf.py:85-94 on the line parent(b) -> b
        worst = max(worst, np.abs(isq[:, b] * vs[:, b] - (pl[:, b] ** 2 + ql[:, b] ** 2)).max())
        worst = max(worst, np.abs(vs[:, par[b]] - 2 * (t.r[b] * pl[:, b] + t.x[b] * ql[:, b])
                                  - (t.r[b] ** 2 + t.x[b] ** 2) * isq[:, b] - vs[:, b]).max())
    assert worst < 1e-11


def test_hip_power_flow_against_the_reference_nlp_statement(net, base_loads):
    """The shortest chain to utils/pf.py this container allows: the HIP solvers (pf_solve_batch through the C ABI, all
    three: tree-Newton, sweeps, dense LU) against oracle/pf_nlp_oracle.ReferenceNLP — the variables, bounds, objective and
    constraints of pf.py:39-94 written out verbatim and solved by SciPy's SLSQP from the start point Pyomo + IPOPT use —
    on the run_pf.py:37-54 inputs, the base case, heavy load, reverse flow and 8 random cases.  (a) voltages, line flows
    and squared currents agree <= 1e-8 (SLSQP's own accuracy; north_star's bar is 1e-6); (b) the HIP result, put into the
    NLP's variables, satisfies every constraint of pf.py:65-94 to 1e-11 — no solver in between."""
    from oracle.pf_nlp_oracle import ReferenceNLP
    from safe_marl_amd.network import build_tables
    p, q = base_loads
    rng = np.random.default_rng(44)
    buses = net["bus_numbers"]
    blds = [buses.index(b) for b in net["buildings"]]
    run_pf_p = np.array([0.0 if net["bus_types"][b] == 1 else 0.1 for b in buses])
    run_pf_p[blds] -= 0.05 + 0.075 - 0.005                                   # Pred, Ppv, -Pesc of run_pf.py:37-54
    run_pf_q = np.array([0.0 if net["bus_types"][b] == 1 else 0.005 for b in buses])
    cases = [("run_pf", run_pf_p, run_pf_q), ("base", p, q), ("heavy", 1.5 * p, 1.5 * q),
             ("reverse", p - 0.25 * (np.arange(len(p)) > 0) * rng.uniform(0.5, 1.0, len(p)), 0.3 * q)]
    for _ in range(8):
        cases.append(("random", p * rng.uniform(0.0, 1.6, len(p)), q * rng.uniform(-0.5, 1.6, len(p))))
    P = np.stack([c[1] for c in cases])
    Q = np.stack([c[2] for c in cases])
    t = build_tables(net)
    lines = list(net["line_connections"])
    bus_of_line = [next(b for b in range(t.n_bus) if t.line_of_bus[b] == l) for l in lines]      # receiving bus of line l
    nlp = [ReferenceNLP(net, P[i], Q[i]) for i in range(len(cases))]
    ref = [m.solve("SLSQP") for m in nlp]
    for i, r in enumerate(ref):
        assert r["residual"] < 1e-10 and r["vm"].min() > 0.8, (cases[i][0], r["residual"])
    for solver in (0, 2, 1):                               # tree Newton, sweeps, dense LU (|V| only)
        out = _solve(net, P, Q, want_branch=(solver != 1), solver=solver)
        assert out["failed"].sum() == 0
        for i, (name, _, _) in enumerate(cases):
            assert np.abs(out["v"][i] - ref[i]["vm"]).max() < 1e-8, (solver, name)
            if solver == 1:
                continue
            pl, ql, isq = out["pl"][i][bus_of_line], out["ql"][i][bus_of_line], out["isqr"][i][bus_of_line]
            assert np.abs(pl - ref[i]["Pl"]).max() < 1e-8 and np.abs(ql - ref[i]["Ql"]).max() < 1e-8, (solver, name)
            assert np.abs(isq - ref[i]["Isqr"]).max() < 1e-8, (solver, name)
            # (b) HIP result in the NLP's variables: [Vsqr (non-slack) | Pl | Ql | Isqr | Ps | Qs]
            m = nlp[i]
            vs = out["v"][i] ** 2
            fr_slack = m.fr == m.slack
            ps = (pl + m.R * isq)[fr_slack].sum() + P[i][m.slack]             # what the slack must supply (pf.py:65-72)
            qs = (ql + m.X * isq)[fr_slack].sum() + Q[i][m.slack]
            x = np.concatenate([vs[m.free_v], pl, ql, isq, [ps, qs]])
            assert np.abs(m.constraints(x)).max() < 1e-11, (solver, name, np.abs(m.constraints(x)).max())
            assert (x >= m.lb).all()


def test_sweep_extrapolation_changes_the_count_not_the_solution(net):
    """csrc/flex_device.h pf_sweep, round 5: the two-sweep extrapolation of the dominant error mode (calibrated at
    flexenv_create from the series) removes sweeps; the fixed point is the same — the Newton verification confirms both at
    1e-12 — so a whole warm-started episode with and without it agrees to 1e-12 in V and rewards, with fewer sweeps."""
    import torch
    from safe_marl_amd.series import make_synthetic_series
    from safe_marl_amd.flex_env import VecFlexProvisionEnv
    s = make_synthetic_series(net, n_days=20)
    n = 512
    envs = [VecFlexProvisionEnv({}, n, series=s, net=net, seed=5, warm_start=True, sweep_accel=on) for on in (True, False)]
    for e in envs:
        e.reset()
    g = torch.Generator(device="cuda").manual_seed(3)
    sweeps = [0.0, 0.0]
    worst_v = worst_r = 0.0
    for t in range(60):
        acts = (0.5 + 0.5 * torch.rand(n, 5, 4, device="cuda", generator=g)).float()
        out = [e.step(acts, obs_rows=True, auto_reset=True) for e in envs]
        worst_r = max(worst_r, (out[0][0] - out[1][0]).abs().max().item())
        worst_v = max(worst_v, (envs[0].peek("V") - envs[1].peek("V")).abs().max().item())
        assert torch.equal(out[0][1], out[1][1])
        for k, e in enumerate(envs):
            sweeps[k] += e.peek("PF_SWEEPS").float().mean().item() / 60
            assert e.peek("PF_ITERS").sum().item() == 0 and e.failed.sum().item() == 0
    assert worst_v < 1e-12 and worst_r < 1e-12, (worst_v, worst_r)
    assert sweeps[0] < sweeps[1] - 0.5, sweeps
