"""GPU parity: the HIP environment (through the C ABI) vs the scalar CPU oracle on identical
injected episodes.  Tolerances: V, E, reward terms <= 1e-10 absolute (north_star: 1e-6);
done/steps/failed exact; observations equal after the float32 cast their consumer applies
(util.py:145)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-10


def _oracle_envs(net, series, n, cfg=None, alg=None):
    from oracle.env_oracle import FlexEnvOracle
    return [FlexEnvOracle(net, cfg or {}, series.active, series.reactive, series.pv, series.price,
                          time_delta=series.time_delta, alg=alg) for _ in range(n)]


def _spec(rng, n, series, na, e_max=0.025):
    day = rng.integers(0, series.n_start_days(96), n).astype(np.int32)
    hour = rng.integers(0, 24, n).astype(np.int32)
    interval = rng.integers(0, series.per_hour, n).astype(np.int32)
    e0 = rng.uniform(0.9 * e_max / 2, 1.1 * e_max / 2, (n, na))
    a0 = rng.uniform(0, 1, (n, 4 * na))
    return dict(day=day, hour=hour, interval=interval, e0=e0, a0=a0)


def _compare_state(vec, oracles, tag):
    v = vec.peek("V").cpu().numpy()
    e = vec.peek("E").cpu().numpy()
    ei = vec.peek("E_INIT").cpu().numpy()
    for i, o in enumerate(oracles):
        assert np.abs(v[i] - o.current_voltage).max() < TOL, tag
        assert np.abs(e[i] - np.array(o.current_ess_energy)).max() < TOL, tag
        assert np.abs(ei[i] - np.array(o.initial_ess_energy)).max() < TOL, tag


@pytest.mark.parametrize("solver,warm", [(0, False), (2, False), (2, True)])
@pytest.mark.parametrize("action_range", [(0.5, 1.0), (0.0, 1.0)])
def test_full_episode_matches_oracle(net, series_small, action_range, solver, warm):
    """95 steps x 48 envs: reward, info, done, V, E, obs, state at every step.  (0.5,1.0) is the range the
    reference actually delivers; (0,1) exercises the ESS clipping branches."""
    import torch
    from safe_marl_amd.flex_env import VecFlexProvisionEnv
    n, na = 47, 5     # odd on purpose: the last wavefront has a spare lane group
    rng = np.random.default_rng(3)
    vec = VecFlexProvisionEnv({}, n, series=series_small, net=net, solver=solver, warm_start=warm)
    oracles = _oracle_envs(net, series_small, n)
    spec = _spec(rng, n, series_small, na)
    obs = vec.reset(spec=spec).cpu().numpy()
    assert vec.failed.sum().item() == 0
    for i, o in enumerate(oracles):
        oo, _ = o.reset(spec=(spec["day"][i], spec["hour"][i], spec["interval"][i], spec["e0"][i], spec["a0"][i]))
        assert np.array_equal(np.stack(oo).astype(np.float32), obs[i])
    _compare_state(vec, oracles, "reset")
    state = vec.get_state().cpu().numpy()
    for i, o in enumerate(oracles):
        assert np.abs(state[i] - o.get_state()).max() < TOL
    for t in range(95):
        acts = rng.uniform(*action_range, (n, na, 4)).astype(np.float32)   # float32 like util.py:184
        reward, done, info = vec.step(torch.from_numpy(acts).cuda())
        obs = vec.get_obs().cpu().numpy()
        reward, done, info = reward.cpu().numpy(), done.cpu().numpy(), info.cpu().numpy()
        for i, o in enumerate(oracles):
            r, d, inf = o.step(acts[i].astype(np.float64))
            assert abs(r - reward[i]) < TOL, (t, i)
            assert d == bool(done[i])
            ref = [inf[k] for k in ("reward", "revenue", "der_cost", "ess_cost", "discomfort_penalty",
                                    "voltage_penalty", "cumulative_reward")]
            assert np.abs(np.array(ref) - info[i]).max() < TOL
            oo = np.stack(o.get_obs()).astype(np.float32)
            assert np.allclose(oo, obs[i], rtol=2e-7, atol=0)
        if t % 10 == 0 or t == 94:
            _compare_state(vec, oracles, f"step {t}")
    assert done.all() and (vec.peek("STEPS").cpu().numpy() == 96).all()
    state = vec.get_state().cpu().numpy()
    for i, o in enumerate(oracles):
        assert np.abs(state[i] - o.get_state()).max() < TOL


@pytest.mark.parametrize("pf_tol,bound", [(1e-6, 1e-7), (1e-8, 1e-9)])
def test_looser_power_flow_tolerances(net, series_small, pf_tol, bound):
    """BASELINE.json's north_star asks for bus voltages and step() rewards within 1e-6 of the CPU reference.  The product
    default iterates the power flow to a mismatch of 1e-12 pu (every other test here: 1e-10 against the oracle); `pf_tol` is
    a constructor argument.  At pf_tol = 1e-6 — `tolerance_sibling` in bench.py's line, 25 % more env-steps/s — a whole
    warm-started episode still sits an order of magnitude inside the contract, and at 1e-8 (the default tolerance of the
    reference's own IPOPT solve, pf.py:101-102) within 1e-9: reward, every info term, voltages and ESS energy against the
    oracle (which solves to 1e-12), done exactly.  At neither tolerance may the sweeps hand an unconverged iterate to the
    Newton verification (round 4: at 1e-9 / 1e-8 the fp32 convergence estimate was accepted on a far anchor, a third to a
    half of the solves then took a Newton step, and the launch 18.5 us instead of 11)."""
    import torch
    from safe_marl_amd.flex_env import VecFlexProvisionEnv
    n, na = 47, 5
    rng = np.random.default_rng(8)
    vec = VecFlexProvisionEnv({}, n, series=series_small, net=net, warm_start=True, pf_tol=pf_tol)
    oracles = _oracle_envs(net, series_small, n)
    spec = _spec(rng, n, series_small, na)
    vec.reset(spec=spec)
    for i, o in enumerate(oracles):
        o.reset(spec=(spec["day"][i], spec["hour"][i], spec["interval"][i], spec["e0"][i], spec["a0"][i]))
    worst_r = worst_v = 0.0
    newton = 0
    for t in range(95):
        acts = rng.uniform(0.0, 1.0, (n, na, 4)).astype(np.float32)
        reward, done, info = vec.step(torch.from_numpy(acts).cuda())
        reward, done, info = reward.cpu().numpy(), done.cpu().numpy(), info.cpu().numpy()
        v = vec.peek("V").cpu().numpy()
        e = vec.peek("E").cpu().numpy()
        newton += int(vec.peek("PF_ITERS").sum().item())
        for i, o in enumerate(oracles):
            r, d, inf = o.step(acts[i].astype(np.float64))
            assert d == bool(done[i])
            ref = [inf[k] for k in ("reward", "revenue", "der_cost", "ess_cost", "discomfort_penalty", "voltage_penalty",
                                    "cumulative_reward")]
            worst_r = max(worst_r, abs(r - reward[i]), np.abs(np.array(ref) - info[i]).max())
            worst_v = max(worst_v, np.abs(np.asarray(o.current_voltage) - v[i]).max(),
                          np.abs(np.asarray(o.current_ess_energy) - e[i]).max())
    assert vec.failed.sum().item() == 0
    assert worst_r < 1e-6 and worst_v < 1e-6, (worst_r, worst_v)          # north_star's tolerance
    assert worst_r < bound and worst_v < bound, (worst_r, worst_v)
    assert newton == 0


def test_fused_obs_equals_separate(net, series_small):
    import torch
    from safe_marl_amd.flex_env import VecFlexProvisionEnv
    n = 32
    rng = np.random.default_rng(5)
    a = VecFlexProvisionEnv({}, n, series=series_small, net=net, seed=11)
    b = VecFlexProvisionEnv({}, n, series=series_small, net=net, seed=11)
    oa = a.reset().clone()
    ob = b.reset().clone()
    assert torch.equal(oa, ob)
    for t in range(30):
        acts = torch.from_numpy(rng.uniform(0.5, 1, (n, 5, 4))).cuda()
        a.step(acts, fuse_obs=True)
        b.step(acts)
        b.get_obs()
        assert torch.equal(a.obs, b.obs)
        assert torch.equal(a.reward, b.reward)


def test_solver_failure_semantics(net, series_small):
    """A8/env:314-337: on a failed solve the reward is computed from the restored previous
    actions/voltages, minus 200; the episode terminates; the data row still advances."""
    import torch
    from safe_marl_amd.flex_env import VecFlexProvisionEnv
    from safe_marl_amd.series import SeriesTable
    from oracle.env_oracle import FlexEnvOracle
    sc = SeriesTable(series_small.table.copy(), series_small.n_bus, series_small.n_agents)
    rng = np.random.default_rng(9)
    spec = _spec(rng, 2, sc, 5)
    spec["day"][:] = 3; spec["hour"][:] = 2
    spec["interval"][:] = [1, 2]
    s1 = 2 + 2 * 4 + 3 * 96
    # step k >= 2 solves with row start+k-1 (A2): env 1 meets the poisoned row at step 3, env 0 at step 4
    sc.table[s1 + 2, 1:33] *= 40.0
    vec = VecFlexProvisionEnv({}, 2, series=sc, net=net)
    vec.reset(spec=spec)
    oracles = [FlexEnvOracle(net, {}, sc.active, sc.reactive, sc.pv, sc.price) for _ in range(2)]
    for i, o in enumerate(oracles):
        o.reset(spec=(spec["day"][i], spec["hour"][i], spec["interval"][i], spec["e0"][i], spec["a0"][i]))
    fails = []
    for t in range(3):
        acts = rng.uniform(0.5, 1, (2, 5, 4))
        reward, done, info = vec.step(torch.from_numpy(acts).cuda())
        failed = vec.failed.cpu().numpy()
        obs = vec.get_obs().cpu().numpy()
        for i, o in enumerate(oracles):
            r, d, inf = o.step(acts[i])
            assert abs(r - reward[i].item()) < TOL
            assert d == bool(done[i].item())
            assert bool(failed[i]) == bool(inf.get("solver_failed", False))
            assert np.allclose(np.stack(o.get_obs()).astype(np.float32), obs[i], rtol=2e-7, atol=0)
        fails.append(failed.copy())
        _compare_state(vec, oracles, f"fail step {t}")
    assert [list(f) for f in fails] == [[0, 0], [0, 0], [0, 1]]
    assert reward[1].item() < -190 and done[1].item() == 1 and done[0].item() == 0


def test_safemaddpg_raw_action_branch(net, series_small):
    """env:268-274: with alg == 'safemaddpg' the actions are physical set-points, not scaled."""
    import torch
    from safe_marl_amd.flex_env import VecFlexProvisionEnv
    n = 8
    rng = np.random.default_rng(13)
    vec = VecFlexProvisionEnv({"alg": "safemaddpg"}, n, series=series_small, net=net)
    oracles = _oracle_envs(net, series_small, n, alg="safemaddpg")
    spec = _spec(rng, n, series_small, 5)
    vec.reset(spec=spec)
    for i, o in enumerate(oracles):
        o.reset(spec=(spec["day"][i], spec["hour"][i], spec["interval"][i], spec["e0"][i], spec["a0"][i]))
    for t in range(10):
        acts = rng.uniform(0.5, 1.0, (n, 5, 4))
        reward, done, _ = vec.step(torch.from_numpy(acts).cuda())
        for i, o in enumerate(oracles):
            r, d, _ = o.step(acts[i])
            assert abs(r - reward[i].item()) < TOL
    _compare_state(vec, oracles, "raw")


def test_philox_reset_stream_matches_restatement(net, series_small):
    """Device-drawn episodes (no injection) equal the oracle's restatement of the Philox stream."""
    from safe_marl_amd.flex_env import VecFlexProvisionEnv
    from oracle.env_oracle import reset_draws, DEFAULT_CFG
    n = 41            # odd on purpose: the last wavefront has a spare lane group
    vec = VecFlexProvisionEnv({}, n, series=series_small, net=net, seed=1234)
    obs = vec.reset().cpu().numpy()
    oracles = _oracle_envs(net, series_small, n)
    start = vec.peek("START").cpu().numpy()
    for i, o in enumerate(oracles):
        spec = reset_draws(i, 0, 1234, 5, series_small.n_start_days(96), series_small.per_hour, DEFAULT_CFG)
        oo, _ = o.reset(spec=spec)
        assert o.start == start[i]
        assert np.allclose(np.stack(oo).astype(np.float32), obs[i], rtol=2e-7, atol=0)
    _compare_state(vec, oracles, "philox reset")
    # masked second reset: only masked envs move on to episode counter 1
    mask = np.zeros(n, np.uint8)
    mask[::3] = 1
    vec.reset(mask=mask)
    ep = vec.peek("EPISODE").cpu().numpy()
    assert (ep == 1 + mask).all()
    start2 = vec.peek("START").cpu().numpy()
    for i in range(n):
        if mask[i]:
            spec = reset_draws(i, 1, 1234, 5, series_small.n_start_days(96), series_small.per_hour, DEFAULT_CFG)
            assert start2[i] == spec[2] + spec[1] * 4 + spec[0] * 96
        else:
            assert start2[i] == start[i]


def test_three_agent_variant(net, series_small):
    """BASELINE.json's '3-agent' config: buildings [5,15,25]."""
    import torch
    from safe_marl_amd.network import create_network
    from safe_marl_amd.series import make_synthetic_series
    from safe_marl_amd.flex_env import VecFlexProvisionEnv
    from oracle.env_oracle import FlexEnvOracle
    args = {"buildings": [5, 15, 25], "pv_nodes": [5, 15, 25], "ess_nodes": [5, 15, 25]}
    net3 = create_network(args)
    s3 = make_synthetic_series(net3, n_days=10)
    n = 8
    rng = np.random.default_rng(17)
    vec = VecFlexProvisionEnv(args, n, series=s3, net=net3)
    spec = _spec(rng, n, s3, 3)
    vec.reset(spec=spec)
    oracles = [FlexEnvOracle(net3, {}, s3.active, s3.reactive, s3.pv, s3.price) for _ in range(n)]
    for i, o in enumerate(oracles):
        o.reset(spec=(spec["day"][i], spec["hour"][i], spec["interval"][i], spec["e0"][i], spec["a0"][i]))
    for t in range(12):
        acts = rng.uniform(0, 1, (n, 3, 4))
        reward, _, _ = vec.step(torch.from_numpy(acts).cuda())
        obs = vec.get_obs().cpu().numpy()
        for i, o in enumerate(oracles):
            r, _, _ = o.step(acts[i])
            assert abs(r - reward[i].item()) < TOL
            assert np.allclose(np.stack(o.get_obs()).astype(np.float32), obs[i], rtol=2e-7, atol=0)
    assert obs.shape == (n, 3, 144)


def test_auto_reset_in_launch_equals_step_then_masked_reset(net, series_small):
    """FLEX_STEP_AUTORESET: the fused restart gives the same state, observations and reset-stream position as
    flexenv_step followed by flexenv_reset(mask=done)."""
    import torch
    from safe_marl_amd.flex_env import VecFlexProvisionEnv
    n = 37
    rng = np.random.default_rng(23)
    args = {"episode_limit": 7}                     # 6 steps per episode: several restarts in a short run
    a = VecFlexProvisionEnv(args, n, series=series_small, net=net, seed=77)
    b = VecFlexProvisionEnv(args, n, series=series_small, net=net, seed=77)
    assert torch.equal(a.reset(), b.reset())
    restarts = 0
    for t in range(20):
        acts = torch.from_numpy(rng.uniform(0.5, 1, (n, 5, 4))).cuda()
        ra, da, ia = a.step(acts, fuse_obs=True, auto_reset=True)
        rb, db, ib = b.step(acts, fuse_obs=True)
        b.reset(mask=db, obs_out=b.obs)
        restarts += int(db.sum().item())
        assert torch.equal(ra, rb) and torch.equal(da, db) and torch.equal(ia, ib)
        assert torch.equal(a.obs, b.obs)
        for k in ("V", "E", "E_INIT", "PRED", "STEPS", "ROW", "START", "EPISODE", "CUMREW"):
            assert torch.equal(a.peek(k), b.peek(k)), (t, k)
    # 20 steps of 6-step episodes: three scheduled restarts per env — and more where an episode ends early: with
    # episode_limit = 7 the step is 24/7 h long and a full discharge takes E_next below zero, which pf.py:45 makes an
    # infeasible NLP (solver_failed; round 5)
    assert restarts >= 3 * n


def test_env_on_a_45_bus_feeder_one_env_per_wavefront():
    """The EPW = 1 kernels (more than 32 PQ buses): reset, step, obs and auto-reset against the oracle."""
    import torch
    from tests.test_pf_gpu import _random_feeder
    from safe_marl_amd.series import make_synthetic_series
    from safe_marl_amd.flex_env import VecFlexProvisionEnv
    from oracle.env_oracle import FlexEnvOracle
    blds = [7, 19, 33, 41]
    netx = _random_feeder(45, 11, blds)
    sx = make_synthetic_series(netx, n_days=6)
    n = 9
    rng = np.random.default_rng(31)
    vec = VecFlexProvisionEnv({"buildings": blds, "pv_nodes": blds, "ess_nodes": blds}, n, series=sx, net=netx)
    spec = _spec(rng, n, sx, 4)
    obs = vec.reset(spec=spec).cpu().numpy()
    oracles = [FlexEnvOracle(netx, {}, sx.active, sx.reactive, sx.pv, sx.price) for _ in range(n)]
    for i, o in enumerate(oracles):
        oo, _ = o.reset(spec=(spec["day"][i], spec["hour"][i], spec["interval"][i], spec["e0"][i], spec["a0"][i]))
        assert np.allclose(np.stack(oo).astype(np.float32), obs[i], rtol=2e-7, atol=0)
    for t in range(12):
        acts = rng.uniform(0, 1, (n, 4, 4))
        reward, done, info = vec.step(torch.from_numpy(acts).cuda(), fuse_obs=True)
        obs = vec.obs.cpu().numpy()
        for i, o in enumerate(oracles):
            r, d, _ = o.step(acts[i])
            assert abs(r - reward[i].item()) < TOL and d == bool(done[i].item())
            assert np.allclose(np.stack(o.get_obs()).astype(np.float32), obs[i], rtol=2e-7, atol=0)
    _compare_state(vec, oracles, "45-bus")
    state = vec.get_state().cpu().numpy()
    for i, o in enumerate(oracles):
        assert np.abs(state[i] - o.get_state()).max() < TOL


def test_full_size_batch_4096_envs_against_c_oracle(net):
    """BASELINE.json config 2 at full size: 4096 envs x one whole episode, every step compared with the C restatement
    (OpenMP over envs), plus the size-independent property that the batch result does not depend on batch position."""
    import torch
    from safe_marl_amd.series import make_synthetic_series
    from safe_marl_amd.flex_env import VecFlexProvisionEnv
    from oracle import c_oracle
    s = make_synthetic_series(net, n_days=40)
    n = 4096
    rng = np.random.default_rng(101)
    spec = _spec(rng, n, s, 5)
    vec = VecFlexProvisionEnv({}, n, series=s, net=net, warm_start=True)
    obs = vec.reset(spec=spec).cpu().numpy()
    cenv = c_oracle.COracleEnv(net, s.table, n)
    start = spec["interval"] + spec["hour"] * 4 + spec["day"] * 96
    cobs = cenv.reset(start, spec["e0"], spec["a0"])
    assert cenv.failed.sum() == 0 and vec.failed.sum().item() == 0
    assert np.allclose(cobs, obs, rtol=2e-7, atol=0)
    worst_r = worst_v = worst_e = 0.0
    for t in range(95):
        acts = rng.uniform(0.0, 1.0, (n, 5, 4)).astype(np.float32)
        reward, done, info = vec.step(torch.from_numpy(acts).cuda(), fuse_obs=True)
        r2, d2, i2 = cenv.step(acts.astype(np.float64))
        worst_r = max(worst_r, np.abs(reward.cpu().numpy() - r2).max())
        assert np.array_equal(done.cpu().numpy(), d2)
        assert np.abs(info.cpu().numpy() - i2).max() < 1e-9
        if t % 8 == 0 or t == 94:
            worst_v = max(worst_v, np.abs(vec.peek("V").cpu().numpy() - cenv.V).max())
            worst_e = max(worst_e, np.abs(vec.peek("E").cpu().numpy() - cenv.E).max())
            assert np.allclose(vec.obs.cpu().numpy(), cenv.obs, rtol=2e-7, atol=0)
    assert worst_r < TOL and worst_v < TOL and worst_e < TOL
    assert done.all()
    # batch-position independence: env i of a 4096-batch equals the same episode run alone
    k = 1234
    one = VecFlexProvisionEnv({}, 1, series=s, net=net, warm_start=True)
    one.reset(spec={key: val[k:k + 1] for key, val in spec.items()})
    vec.reset(spec=spec)
    rng2 = np.random.default_rng(7)
    for t in range(10):
        acts = rng2.uniform(0.0, 1.0, (n, 5, 4)).astype(np.float32)
        r_all, _, _ = vec.step(torch.from_numpy(acts).cuda())
        r_one, _, _ = one.step(torch.from_numpy(acts[k:k + 1]).cuda())
        assert r_all[k].item() == r_one[0].item()
    assert torch.equal(vec.peek("V")[k], one.peek("V")[0])


def _philox_specs(n, episode, seed, series, na=5):
    """reset_draws (the Philox restatement of oracle/env_oracle.py) for envs 0..n-1 of one episode counter"""
    from oracle.env_oracle import reset_draws, DEFAULT_CFG
    day, hour, interval = np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(n, np.int32)
    e0, a0 = np.zeros((n, na)), np.zeros((n, 4 * na))
    for i in range(n):
        d, h, iv, e, a = reset_draws(i, episode, seed, na, series.n_start_days(96), series.per_hour, DEFAULT_CFG)
        day[i], hour[i], interval[i], e0[i], a0[i] = d, h, iv, e, a
    return dict(day=day, hour=hour, interval=interval, e0=e0, a0=a0)


def test_bench_workload_auto_reset_4096_envs_against_c_oracle_across_episode_boundaries(net):
    """The EXACT workload bench.py times (BASELINE.json config 2 as measured): 4096 envs, warm-started solver,
    ``step(obs_rows=True, auto_reset=True)`` (get_obs() as a row push; the stacked observation is read back through
    ``obs_view()``) — every environment terminates at vector steps 95 and 190 and restarts INSIDE
    the launch from the device's Philox reset stream (model.py:208,255-262: reset per episode).  205 steps = two episode
    boundaries; the C oracle is restarted at each boundary from the restated stream (episode counters 1 and 2).  Reward,
    info, V, E <= 1e-10; done exact; the fused observation (the new episode's first one at a boundary) equal after the
    fp32 cast."""
    import torch
    from safe_marl_amd.series import make_synthetic_series
    from safe_marl_amd.flex_env import VecFlexProvisionEnv
    from oracle import c_oracle
    s = make_synthetic_series(net, n_days=40)
    n, seed = 4096, 1234
    rng = np.random.default_rng(202)
    vec = VecFlexProvisionEnv({}, n, series=s, net=net, warm_start=True, seed=seed)
    obs = vec.reset().cpu().numpy()                                   # device-drawn: episode counter 0
    cenv = c_oracle.COracleEnv(net, s.table, n)
    spec = _philox_specs(n, 0, seed, s)
    cobs = cenv.reset(spec["interval"] + spec["hour"] * 4 + spec["day"] * 96, spec["e0"], spec["a0"])
    assert cenv.failed.sum() == 0 and vec.failed.sum().item() == 0
    assert np.allclose(cobs, obs, rtol=2e-7, atol=0)
    worst_r = worst_v = worst_e = worst_i = 0.0
    boundaries = 0
    for t in range(205):
        acts = rng.uniform(0.5, 1.0, (n, 5, 4)).astype(np.float32)    # the range bench.py draws from
        reward, done, info = vec.step(torch.from_numpy(acts).cuda(), obs_rows=True, auto_reset=True)    # bench.py's one_step
        r2, d2, i2 = cenv.step(acts.astype(np.float64))
        worst_r = max(worst_r, np.abs(reward.cpu().numpy() - r2).max())
        worst_i = max(worst_i, np.abs(info.cpu().numpy() - i2).max())
        assert np.array_equal(done.cpu().numpy(), d2), t
        assert vec.failed.sum().item() == 0
        if d2.all():                                                   # an episode boundary for the whole batch
            boundaries += 1
            spec = _philox_specs(n, boundaries, seed, s)
            cenv.reset(spec["interval"] + spec["hour"] * 4 + spec["day"] * 96, spec["e0"], spec["a0"])
            assert cenv.failed.sum() == 0
            assert (vec.peek("EPISODE").cpu().numpy() == boundaries + 1).all()
            assert np.array_equal(vec.peek("START").cpu().numpy(), cenv.start)
        else:
            assert not d2.any()
        if t % 8 == 0 or d2.all() or t in (95, 96, 190, 191, 204):
            worst_v = max(worst_v, np.abs(vec.peek("V").cpu().numpy() - cenv.V).max())
            worst_e = max(worst_e, np.abs(vec.peek("E").cpu().numpy() - cenv.E).max())
            assert np.allclose(vec.obs_view().cpu().numpy(), cenv.obs, rtol=2e-7, atol=0), t
    assert boundaries == 2
    assert worst_r < TOL and worst_v < TOL and worst_e < TOL and worst_i < 1e-9, (worst_r, worst_v, worst_e, worst_i)


@pytest.mark.parametrize("cfg,blds,n", [({}, [5, 10, 15, 20, 25], 37), ({"episode_limit": 7}, [5, 15, 25], 64),
                                        ({"history": 3, "episode_limit": 9}, [5, 10, 15, 20, 25], 33),
                                        ({"history": 30, "episode_limit": 40}, [2, 6, 12, 18, 22, 25, 30, 33], 9)])
def test_row_push_leaves_the_observation_a_stacked_copy_would(net, cfg, blds, n):
    """FLEX_STEP_OBS_ROWS (include/flexenv.h): get_obs() as a push of the step's feature row into the env's mirror ring +
    ``obs_view()`` == get_obs() as a stacked copy (``fuse_obs``), bit for bit, step by step, across in-launch restarts, for
    history / agent counts on both sides of the register path; and the two forms can be mixed on one environment."""
    import torch
    from safe_marl_amd.network import create_network
    from safe_marl_amd.series import make_synthetic_series
    from safe_marl_amd.flex_env import VecFlexProvisionEnv
    args = dict(cfg, buildings=blds, pv_nodes=blds, ess_nodes=blds)
    netx = create_network(args)
    sx = make_synthetic_series(netx, n_days=8)
    a = VecFlexProvisionEnv(args, n, series=sx, net=netx, seed=7, warm_start=True)
    b = VecFlexProvisionEnv(args, n, series=sx, net=netx, seed=7, warm_start=True)
    c = VecFlexProvisionEnv(args, n, series=sx, net=netx, seed=7, warm_start=True)
    g = torch.Generator(device="cuda").manual_seed(3)
    first = a.reset().clone()
    b.reset(want_obs=True)
    c.reset()
    assert torch.equal(first, b.obs_view().clone()) and torch.equal(first, c.obs)
    steps = 3 * int(cfg.get("episode_limit", 96)) + 5 if "episode_limit" in cfg else 60
    for t in range(steps):
        acts = 0.5 + 0.5 * torch.rand(n, len(blds), 4, device="cuda", generator=g)
        ra, da, _ = a.step(acts, fuse_obs=True, auto_reset=True)
        rb, db, _ = b.step(acts, obs_rows=True, auto_reset=True)
        (c.step(acts, fuse_obs=True, auto_reset=True) if t % 3 else c.step(acts, obs_rows=True, auto_reset=True))
        assert torch.equal(ra, rb) and torch.equal(da, db)
        want = a.obs.clone()
        assert torch.equal(b.obs_view(), want), t
        assert torch.equal(c.obs if t % 3 else c.obs_view(), want), t
    assert int(da.sum()) == 0 or "episode_limit" in cfg


@pytest.mark.parametrize("cfg,blds", [
    ({"history": 1}, [5, 10, 15, 20, 25]),                                            # no stacking (env:387)
    ({"history": 3, "episode_limit": 12}, [5, 10, 15, 20, 25]),
    ({"eta_ch": 0.95, "eta_dis": 0.85, "e_max": 0.03, "e_min": 0.002, "p_ch_max": 0.008, "p_dis_max": 0.004,
      "v_min": 0.95, "v_max": 1.02, "cos_phi_max": 0.9, "max_power_reduction": 0.3, "pv_cost": 0.08, "ess_cost": 0.01,
      "discomfort_coeff": 0.4, "voltage_coeff": 2.5}, [5, 10, 15, 20, 25]),           # every yaml scalar moved
    ({}, [18]),                                                                       # one agent, at the feeder's end
    ({"history": 30}, [2, 6, 12, 18, 22, 25, 30, 33]),                                # 8 agents x 30 history: beyond the register path
])
def test_non_default_configurations(net, cfg, blds):
    import torch
    from safe_marl_amd.network import create_network
    from safe_marl_amd.series import make_synthetic_series
    from safe_marl_amd.flex_env import VecFlexProvisionEnv
    from oracle.env_oracle import FlexEnvOracle
    args = dict(cfg, buildings=blds, pv_nodes=blds, ess_nodes=blds)
    netx = create_network(args)
    sx = make_synthetic_series(netx, n_days=8)
    na, n = len(blds), 11
    rng = np.random.default_rng(41)
    vec = VecFlexProvisionEnv(args, n, series=sx, net=netx)
    H = cfg.get("history", 24)
    assert vec.obs_size == 6 * H and vec.obs.shape == (n, na, 6 * H)
    spec = _spec(rng, n, sx, na, e_max=cfg.get("e_max", 0.025))
    spec["day"] = rng.integers(0, sx.n_start_days(cfg.get("episode_limit", 96)), n).astype(np.int32)
    obs = vec.reset(spec=spec).cpu().numpy()
    oracles = [FlexEnvOracle(netx, cfg, sx.active, sx.reactive, sx.pv, sx.price) for _ in range(n)]
    for i, o in enumerate(oracles):
        oo, _ = o.reset(spec=(spec["day"][i], spec["hour"][i], spec["interval"][i], spec["e0"][i], spec["a0"][i]))
        assert np.allclose(np.stack(oo).astype(np.float32), obs[i], rtol=2e-7, atol=0)
    vp_seen = 0.0
    for t in range(14):
        acts = rng.uniform(0, 1, (n, na, 4))
        reward, done, info = vec.step(torch.from_numpy(acts).cuda(), fuse_obs=(t % 2 == 0))
        obs = (vec.obs if t % 2 == 0 else vec.get_obs()).cpu().numpy()
        info = info.cpu().numpy()
        for i, o in enumerate(oracles):
            r, d, inf = o.step(acts[i])
            assert abs(r - reward[i].item()) < TOL and d == bool(done[i].item())
            assert abs(inf["voltage_penalty"] - info[i, 5]) < TOL and abs(inf["der_cost"] - info[i, 2]) < TOL
            assert np.allclose(np.stack(o.get_obs()).astype(np.float32), obs[i], rtol=2e-7, atol=0)
            vp_seen = max(vp_seen, inf["voltage_penalty"])
    _compare_state(vec, oracles, str(cfg))
    if "v_min" in cfg:
        assert vp_seen > 0          # the tightened band really produced penalties


def test_looser_power_flow_tolerance_stays_inside_the_parity_bar(net, series_small):
    """north_star asks for voltages and rewards within 1e-6 of the CPU reference; the default solver threshold (1e-12)
    is six orders tighter than that.  At 1e-8 — the accuracy class of the reference's own NLP solve — a 95-step episode
    still agrees with the (1e-12) oracle to 1e-7, i.e. the bar leaves a decade of room."""
    import torch
    from safe_marl_amd.flex_env import VecFlexProvisionEnv
    n, na = 32, 5
    rng = np.random.default_rng(5)
    vec = VecFlexProvisionEnv({}, n, series=series_small, net=net, warm_start=True, pf_tol=1e-8)
    oracles = _oracle_envs(net, series_small, n)
    spec = _spec(rng, n, series_small, na)
    vec.reset(spec=spec)
    for i, o in enumerate(oracles):
        o.reset(spec=(spec["day"][i], spec["hour"][i], spec["interval"][i], spec["e0"][i], spec["a0"][i]))
    worst_r = worst_v = 0.0
    for t in range(95):
        acts = rng.uniform(0.5, 1.0, (n, na, 4)).astype(np.float32)
        reward, done, info = vec.step(torch.from_numpy(acts).cuda())
        reward = reward.cpu().numpy()
        v = vec.peek("V").cpu().numpy()
        for i, o in enumerate(oracles):
            r, _, _ = o.step(acts[i].astype(np.float64))
            worst_r = max(worst_r, abs(r - reward[i]))
            worst_v = max(worst_v, np.abs(v[i] - o.current_voltage).max())
    assert worst_r < 1e-7 and worst_v < 1e-7, (worst_r, worst_v)


@pytest.mark.parametrize("solver", [0, 2])
def test_e_next_outside_its_declared_domain_is_a_solver_failure(net, series_small, solver):
    """pf.py:41-45 (E_next in NonNegativeReals, pinned by pf.py:96-98): the HIP step and reset take the reference's failure
    path exactly where the oracle does — e_min = -0.01 and full discharge reach E_next < 0 at the third step; an injected
    reset whose first solve would leave E_next < 0 reports failed (env:150-153 with nothing left to re-draw)."""
    import torch
    from safe_marl_amd.flex_env import VecFlexProvisionEnv
    cfg = {"e_min": -0.01, "p_dis_max": 0.05}
    n, na = 3, 5
    rng = np.random.default_rng(5)
    spec = _spec(rng, n, series_small, na)
    spec["e0"][:] = 0.0125
    spec["e0"][2] = 0.0002                                  # env 2: already the reset's solve leaves E_next < 0 ...
    spec["a0"] = np.tile([0.5, 0.0, 0.0, 0.5], (n, na))
    spec["a0"][2] = np.tile([0.5, 0.0, 1.0, 0.5], na)       # ... because its initial action discharges
    vec = VecFlexProvisionEnv(cfg, n, series=series_small, net=net, solver=solver)
    oracles = _oracle_envs(net, series_small, n, cfg=cfg)
    vec.reset(spec=spec)
    assert vec.failed.cpu().numpy().tolist() == [0, 0, 1]
    from oracle import pf_oracle
    with pytest.raises(pf_oracle.SolverFailed):
        oracles[2].reset(spec=(spec["day"][2], spec["hour"][2], spec["interval"][2], spec["e0"][2], spec["a0"][2]))
    for i in range(2):
        oracles[i].reset(spec=(spec["day"][i], spec["hour"][i], spec["interval"][i], spec["e0"][i], spec["a0"][i]))
    acts = np.tile([0.5, 0.0, 1.0, 0.5], (n, na, 1))
    acts[1, :, 2] = 0.0                                     # env 1 never discharges: stays feasible
    seen = []
    for t in range(3):
        reward, done, info = vec.step(torch.from_numpy(acts).cuda())
        failed = vec.failed.cpu().numpy()
        obs = vec.get_obs().cpu().numpy()
        for i in range(2):
            r, d, inf = oracles[i].step(acts[i])
            assert abs(r - reward[i].item()) < TOL and d == bool(done[i].item())
            assert bool(failed[i]) == bool(inf.get("solver_failed", False))
            assert np.allclose(np.stack(oracles[i].get_obs()).astype(np.float32), obs[i], rtol=2e-7, atol=0)
        seen.append(failed[:2].tolist())
        _compare_state(vec, oracles[:2], f"domain step {t}")
    assert seen == [[0, 0], [0, 0], [1, 0]]
    assert reward[0].item() < -190 and done[0].item() == 1 and done[1].item() == 0


def test_pushes_the_row_ring_would_not_see_are_refused(net, series_small):
    """while a row ring is registered (flexenv_set_obs_ring) a masked flexenv_reset or a stand-alone flexenv_obs
    would move an environment's history without filing the record the replay's window gather relies on -> FLEX_EINVAL;
    a full reset stays allowed.  The ABI-1 value of the ring flag (2) is refused by flexenv_step."""
    import torch
    from safe_marl_amd import _lib
    from safe_marl_amd.flex_env import VecFlexProvisionEnv, _ptr, _stream
    n = 8
    vec = VecFlexProvisionEnv({}, n, series=series_small, net=net, seed=1)
    vec.reset()
    cursor = torch.zeros(2, dtype=torch.int64, device="cuda")
    ring = torch.zeros(4, n, 5, 8, device="cuda")
    vec.set_obs_ring(cursor, n * 5 * 8, 4)
    mask = torch.ones(n, dtype=torch.uint8, device="cuda")
    with pytest.raises(_lib.FlexLibraryError):
        vec.reset(mask=mask)
    with pytest.raises(_lib.FlexLibraryError):
        vec.get_obs()
    vec.reset()                                              # full reset: allowed
    acts = torch.full((n, 5, 4), 0.75, device="cuda")
    vec.step(acts, obs_ring=ring)                            # and the ring form itself works
    torch.cuda.synchronize()
    assert ring[1].abs().sum().item() > 0
    vec.set_obs_ring(None, 0, 0)
    vec.reset(mask=mask)
    rc = vec.lib.flexenv_step(vec.handle, _ptr(acts), _lib.FLEX_F32, _ptr(vec.reward), _ptr(vec.done), None, None,
                              _ptr(ring), _lib.FLEX_F32, 2, _stream())
    assert rc == _lib.FLEX_EINVAL
    assert vec.lib.flexenv_abi_version() == _lib.FLEX_ABI_VERSION == 2
