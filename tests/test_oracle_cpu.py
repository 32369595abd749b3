"""CPU suite: pins the oracle itself (no GPU).  The reference has no tests, golden vectors or
runnable solver for this path, so the oracle is pinned by
  (i)  two independent algorithms agreeing (NR on the Ybus vs DistFlow backward/forward sweep),
  (ii) zero residual in the reference's own constraint expressions pf.py:65-94,
  (iii) the literature band of the Baran-Wu base case,
  (iv) the C restatement agreeing with the Python one."""
import numpy as np
import pytest

from oracle import pf_oracle
from oracle.env_oracle import FlexEnvOracle, philox4x32_10


def test_base_case_literature_band(net, base_loads):
    p, q = base_loads
    assert abs(p.sum() * 1000 - 3715) < 1e-9 and abs(q.sum() * 1000 - 2300) < 1e-9
    sol = pf_oracle.solve_pf(net, p, q)
    assert abs(sol["vm"].min() - 0.9131) < 2e-4 and net["bus_numbers"][int(sol["vm"].argmin())] == 18
    loss = sum(net["line_resistances"][k] * sol["Isqr"][k] for k in sol["Isqr"]) * 1000
    assert abs(loss - 202.68) < 0.1


def test_two_algorithms_agree(net, base_loads):
    p, q = base_loads
    rng = np.random.default_rng(0)
    worst = 0.0
    for _ in range(300):
        P = p * rng.uniform(0.0, 1.6, len(p)) - rng.uniform(0, 0.15, len(p)) * (rng.random(len(p)) < 0.2)
        Q = q * rng.uniform(-0.5, 1.6, len(p))
        P[0] = Q[0] = 0
        vm = pf_oracle.nr_polar(net, P, Q)[0]
        sw = pf_oracle.distflow_sweep(net, P, Q)
        worst = max(worst, np.abs(vm - np.sqrt(sw["Vsqr"])).max())
    assert worst < 1e-10


def test_reference_constraints_are_satisfied(net, base_loads):
    p, q = base_loads
    rng = np.random.default_rng(1)
    buses = net["bus_numbers"]
    for _ in range(50):
        P = p * rng.uniform(0.2, 1.5, len(p))
        Q = q * rng.uniform(0.2, 1.5, len(p))
        sol = pf_oracle.solve_pf(net, P, Q)
        Vs = {b: sol["vm"][i] ** 2 for i, b in enumerate(buses)}
        res = pf_oracle.distflow_residuals(net, dict(zip(buses, P)), dict(zip(buses, Q)), Vs, sol["Pl"], sol["Ql"], sol["Isqr"])
        assert res < 1e-11


def test_run_pf_inputs(net):
    """run_pf.py:37-54: all PQ buses P=0.1, Q=0.005 pu; buildings reduce by 50 %, PV 0.5*pv_cap, ESS charge at
    p_ch_max from e_max/2.  Outputs were never recorded by the reference; check the dict shape, the ESS update
    pf.py:96-98 and self-consistency with the DistFlow constraints."""
    buses = net["bus_numbers"]
    pd = {b: 0 if net["bus_types"][b] == 1 else 0.1 for b in buses}
    qd = {b: 0 if net["bus_types"][b] == 1 else 0.005 for b in buses}
    blds = net["buildings"]
    pred = {b: pd[b] * 0.5 for b in blds}
    ppv = {b: 0.5 * 0.15 for b in blds}
    qpv = {b: 0 for b in blds}
    ch = {b: 0.005 for b in blds}
    dis = {b: 0 for b in blds}
    e0 = {b: 0.025 / 2 for b in blds}
    out = pf_oracle.power_flow_solver(net, pd, qd, pred, ppv, qpv, ch, dis, e0)
    assert set(out) == {"Voltages", "Currents", "Power Flows", "Next ESS Energy"}
    assert len(out["Voltages"]) == 33 and len(out["Currents"]) == 32
    assert out["Voltages"][1] == pytest.approx(1.0, abs=1e-15)
    for b in blds:
        assert out["Next ESS Energy"][b] == pytest.approx(0.0125 + 0.25 * 0.9 * 0.005, abs=1e-15)
    assert 0.80 < min(out["Voltages"].values()) < 1.0


def test_voltage_collapse_raises(net, base_loads):
    p, q = base_loads
    with pytest.raises(pf_oracle.SolverFailed):
        pf_oracle.nr_polar(net, 20 * p, 20 * q)


def test_philox_known_answer():
    """Random123 known-answer vectors for philox4x32-10."""
    assert philox4x32_10((0, 0, 0, 0), (0, 0)) == (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)
    assert philox4x32_10((0xffffffff,) * 4, (0xffffffff,) * 2) == (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)
    assert philox4x32_10((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0)) == \
        (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)


def test_env_quirks(net, series_small):
    """Executable statement of SURVEY.md App. A quirks A2, A3, A5, A9, A16."""
    s = series_small
    env = FlexEnvOracle(net, {}, s.active, s.reactive, s.pv, s.price)
    rng = np.random.default_rng(2)
    e0 = rng.uniform(0.01125, 0.01375, 5)
    a0 = rng.uniform(0, 1, 20)
    obs, state = env.reset(spec=(2, 3, 1, e0, a0))
    start = 1 + 3 * 4 + 2 * 96
    assert env.start == start and env.steps == 1
    assert np.array_equal(env.cur_pd, s.active[start + 1])                  # A2: row 1 at reset
    assert state.shape == (110,) and len(obs) == 5 and obs[0].shape == (144,)
    assert np.all(obs[0][:138] == 0) and obs[0][138] == s.active[start + 1][4]   # A16 zero left-padding
    assert list(env.initial_ess_energy) == list(e0)                         # A5
    assert env.current_ess_energy != list(e0)
    rows = []
    cum = 0.0
    for k in range(1, 96):
        r, d, info = env.step(rng.uniform(0.5, 1, 20))
        assert info["cumulative_reward"] == pytest.approx(cum)              # A9
        cum += r
        rows.append(int(np.where((s.active == env.cur_pd).all(1))[0][0]) - start)
        assert d == (k == 95)                                               # A3: 95 steps
        env.get_obs()
    assert rows[:3] == [1, 2, 3] and rows[-1] == 95                         # A2: after step k the row is k


def test_c_restatement_matches_python(net, series_small, base_loads):
    from oracle import c_oracle
    p, q = base_loads
    rng = np.random.default_rng(4)
    P = p[None] * rng.uniform(0, 1.6, (64, len(p)))
    Q = q[None] * rng.uniform(-0.5, 1.6, (64, len(p)))
    vm, iters = c_oracle.pf_batch(net, P, Q)
    assert (iters >= 0).all()
    for i in range(0, 64, 8):
        assert np.abs(vm[i] - pf_oracle.nr_polar(net, P[i], Q[i])[0]).max() < 1e-11
    s = series_small
    n = 6
    cenv = c_oracle.COracleEnv(net, s.table, n)
    day = rng.integers(0, s.n_start_days(96), n); hour = rng.integers(0, 24, n); itv = rng.integers(0, 4, n)
    start = itv + hour * 4 + day * 96
    e0 = rng.uniform(0.01125, 0.01375, (n, 5)); a0 = rng.uniform(0, 1, (n, 20))
    cobs = cenv.reset(start, e0, a0).copy()
    envs = [FlexEnvOracle(net, {}, s.active, s.reactive, s.pv, s.price) for _ in range(n)]
    for i, e in enumerate(envs):
        oo, _ = e.reset(spec=(day[i], hour[i], itv[i], e0[i], a0[i]))
        assert np.allclose(np.stack(oo).astype(np.float32), cobs[i], rtol=2e-7, atol=0)
    for t in range(95):
        acts = rng.uniform(0, 1, (n, 5, 4))
        r, d, info = cenv.step(acts)
        for i, e in enumerate(envs):
            rr, dd, inf = e.step(acts[i])
            assert abs(rr - r[i]) < 1e-11 and dd == bool(d[i])
            assert abs(inf["cumulative_reward"] - info[i, 6]) < 1e-10
            assert np.allclose(np.stack(e.get_obs()).astype(np.float32), cenv.obs[i], rtol=2e-7, atol=0)
        assert np.abs(cenv.V - np.stack([e.current_voltage for e in envs])).max() < 1e-11
        assert np.abs(cenv.E - np.stack([e.current_ess_energy for e in envs])).max() < 1e-13


def test_reference_nlp_from_ipopts_default_start_lands_on_the_oracle_voltages(net, base_loads):
    """oracle/pf_nlp_oracle.py: the reference's NLP (pf.py:39-94, verbatim) solved by SciPy's SLSQP from the start point
    Pyomo + IPOPT would use (variables without initial values: 0, pushed 1e-2 inside the bounds) — base case, the
    run_pf.py:37-54 injections, heavy load, reverse flow — reaches the oracle's voltages (the fixed point the HIP kernels
    converge to), not a low-voltage root; trust-constr (a trust-region interior point, IPOPT's family) agrees on the base
    case."""
    from oracle.pf_nlp_oracle import ReferenceNLP
    p, q = base_loads
    rng = np.random.default_rng(4)
    buses = net["bus_numbers"]
    blds = [buses.index(b) for b in net["buildings"]]
    run_pf_p = np.array([0.0 if net["bus_types"][b] == 1 else 0.1 for b in buses])
    run_pf_p[blds] -= 0.05 + 0.075 - 0.005                                   # Pred, Ppv, -Pesc of run_pf.py:37-54
    run_pf_q = np.array([0.0 if net["bus_types"][b] == 1 else 0.005 for b in buses])
    cases = [("base", p, q), ("run_pf", run_pf_p, run_pf_q), ("heavy", 1.5 * p, 1.5 * q),
             ("reverse", p - 0.25 * (np.arange(len(p)) > 0) * rng.uniform(0.5, 1.0, len(p)), 0.3 * q)]
    for _ in range(6):
        cases.append(("random", p * rng.uniform(0.0, 1.6, len(p)), q * rng.uniform(-0.5, 1.6, len(p))))
    for name, P, Q in cases:
        ref = pf_oracle.solve_pf(net, P, Q)
        got = ReferenceNLP(net, P, Q).solve("SLSQP")
        assert got["residual"] < 1e-10, (name, got["residual"])          # (SLSQP's own stop flag is not the criterion: feasibility is)
        assert np.abs(got["vm"] - ref["vm"]).max() < 1e-9, (name, np.abs(got["vm"] - ref["vm"]).max())
        assert got["vm"].min() > 0.8                                         # the high-voltage root
    tc = ReferenceNLP(net, p, q).solve("trust-constr")
    assert np.abs(tc["vm"] - pf_oracle.solve_pf(net, p, q)["vm"]).max() < 1e-8


def test_e_next_outside_its_declared_domain_is_a_solver_failure(net, series_small):
    """pf.py:41-45: E_next is a NonNegativeReals variable pinned by the equality pf.py:96-98; once it would be negative the
    reference's NLP is infeasible, pf.py:104-105 raises and env:314-337 takes the failure path (VERDICT r04 missing #3).
    Unreachable at the default YAML (the no-dt clip of env:634 keeps E >= e_min = 0); with e_min = -0.01 full discharge
    walks E_next = E - dt (E - e_min) below zero at the third step.  Python and C oracles agree on when and how."""
    from oracle import c_oracle
    s = series_small
    cfg = {"e_min": -0.01, "p_dis_max": 0.05}
    e0 = np.full((1, 5), 0.0125)
    a0 = np.tile([0.5, 0.0, 0.0, 0.5], 5)[None]
    env = FlexEnvOracle(net, cfg, s.active, s.reactive, s.pv, s.price)
    env.reset(spec=(3, 2, 1, e0[0], a0[0]))
    cenv = c_oracle.COracleEnv(net, s.table, 1, cfg=cfg)
    cenv.reset(np.array([1 + 2 * 4 + 3 * 96]), e0, a0)
    act = np.tile([0.5, 0.0, 1.0, 0.5], (5, 1))            # discharge as hard as the clip allows
    seen = []
    for t in range(3):
        e_before = list(env.current_ess_energy)
        r, d, inf = env.step(act)
        cr, cd, cinfo = cenv.step(act[None])
        failed = bool(inf.get("solver_failed", False))
        seen.append(failed)
        assert abs(cr[0] - r) < 1e-11 and bool(cd[0]) == d and bool(cenv.failed[0]) == failed
        assert np.abs(cenv.E[0] - np.array(env.current_ess_energy)).max() < 1e-13
        if failed:
            assert r < -190 and d and list(env.current_ess_energy) == e_before      # env:319-328: state rolled back
    assert seen == [False, False, True]
    # inside IPOPT's default bound relaxation the NLP stays feasible (pf_oracle.DOMAIN_EPS); beyond it, it does not
    for below, expect_fail in ((5e-9, False), (2e-8, True)):
        env2 = FlexEnvOracle(net, cfg, s.active, s.reactive, s.pv, s.price)
        env2.reset(spec=(3, 2, 1, e0[0], a0[0]))
        dis = 0.02 * cfg["p_dis_max"]
        start = 0.25 * dis / env2.cfg["eta_dis"] - below           # E_next = -below
        env2.initial_ess_energy = [start] * 5
        env2.current_ess_energy = [start] * 5
        _, _, inf2 = env2.step(np.tile([0.5, 0.0, 0.02, 0.5], (5, 1)))
        assert bool(inf2.get("solver_failed", False)) == expect_fail


def test_c_distflow_sweep_agrees_with_the_dense_newton(net, series_small, base_loads):
    """oracle/flexenv_oracle.c pf_solver = 1 (DistFlow backward/forward sweep in the reference's own variables, pf.py:65-94;
    the O(n) CPU algorithm bench.py times beside the dense Newton-Raphson): same voltages as the dense polar NR and as
    oracle/pf_oracle.py on random loadings incl. reverse flow; a whole episode of the env on either solver agrees."""
    from oracle import c_oracle
    p, q = base_loads
    rng = np.random.default_rng(11)
    P = p[None] * rng.uniform(-0.5, 1.6, (64, len(p)))
    Q = q[None] * rng.uniform(-0.5, 1.6, (64, len(p)))
    v_nr, it_nr = c_oracle.pf_batch(net, P, Q)
    v_sw, it_sw = c_oracle.pf_batch(net, P, Q, solver="distflow_sweep")
    assert (it_nr >= 0).all() and (it_sw > 0).all()
    assert np.abs(v_nr - v_sw).max() < 1e-11
    assert np.abs(v_sw[5] - pf_oracle.nr_polar(net, P[5], Q[5])[0]).max() < 1e-11
    # collapse is a failure on both
    _, it_bad = c_oracle.pf_batch(net, 40 * P[:2], 40 * Q[:2], solver="distflow_sweep")
    assert (it_bad < 0).all()
    s, n = series_small, 5
    a = c_oracle.COracleEnv(net, s.table, n)
    b = c_oracle.COracleEnv(net, s.table, n, solver="distflow_sweep")
    start = rng.integers(0, 4, n) + rng.integers(0, 24, n) * 4 + rng.integers(0, s.n_start_days(96), n) * 96
    e0 = rng.uniform(0.01125, 0.01375, (n, 5)); a0 = rng.uniform(0, 1, (n, 20))
    assert np.allclose(a.reset(start, e0, a0), b.reset(start, e0, a0), rtol=2e-7, atol=0)
    for t in range(95):
        acts = rng.uniform(0, 1, (n, 5, 4))
        ra, da, _ = a.step(acts)
        rb, db, _ = b.step(acts)
        assert np.abs(ra - rb).max() < 1e-11 and (da == db).all()
    assert np.abs(a.V - b.V).max() < 1e-11
