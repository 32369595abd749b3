"""GPU parity: the HIP safety projection (through the C ABI) vs the separable CPU oracle, 1e-12 absolute."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("limits", [(0.9, 1.1), (2.0, 2.05), (2.08, 2.3), (0.0, 5.0)])
def test_safety_projection_matches_oracle(net, series_small, limits):
    import torch
    from oracle import safety_oracle
    from oracle.env_oracle import FlexEnvOracle
    from safe_marl_amd.flex_env import VecFlexProvisionEnv
    from safe_marl_amd import safety_signal as ss
    vp = ss.fit_voltage_predictor(net, num_scenarios=300)
    sp, sq, beta = vp.building_terms(net)
    n = 64
    rng = np.random.default_rng(21)
    vec = VecFlexProvisionEnv({"alg": "safemaddpg"}, n, series=series_small, net=net, seed=5)
    vec.reset()
    # take a few raw-action steps so that E differs between envs
    for t in range(3):
        vec.step(torch.from_numpy(rng.uniform(0, 1, (n, 5, 4)) * [0.5, 0.005, 0.005, 0.02]).cuda())
    proposed = rng.uniform(-0.2, 1.2, (n, 5, 4)).astype(np.float32)
    adj, hit = vec.safety_project(torch.from_numpy(proposed).cuda(), sp, sq, beta, *limits)
    adj, hit = adj.cpu().numpy(), hit.cpu().numpy()
    row = vec.peek("ROW").cpu().numpy()
    E = vec.peek("E").cpu().numpy()
    helper = FlexEnvOracle(net, {}, series_small.active, series_small.reactive, series_small.pv, series_small.price)
    buses = list(net["bus_numbers"])
    idx = [buses.index(b) for b in net["buildings"]]
    worst = 0.0
    for i in range(n):
        helper.start = 0
        helper._load_row(int(row[i]))
        pct, _, ch, dis, q = helper._parse(proposed[i].astype(np.float64).reshape(-1), E[i], scaled=True)  # safemaddpg.py:142-174
        changed = False
        for k in range(5):
            x0 = [pct[k], ch[k], dis[k], q[k]]
            x = safety_oracle.solve_separable(x0, helper.cur_pd[idx[k]], helper.cur_qd[idx[k]], sp[k], sq[k], beta[k], *limits)
            got = adj[i, [k, 5 + k, 10 + k, 15 + k]]                    # type-major, safemaddpg.py:297
            worst = max(worst, np.abs(got - x).max())
            changed |= bool(np.abs(np.asarray(x) - np.asarray(x0)).max() > 0)
        assert bool(hit[i]) == changed
    assert worst < 1e-12


def test_fit_voltage_predictor_on_gpu_equals_cpu_fit(net):
    from oracle import pf_oracle
    from safe_marl_amd import safety_signal as ss
    vp = ss.fit_voltage_predictor(net, num_scenarios=120, seed=4)
    P, Q = ss.draw_scenarios(net, 120, 0.3, np.random.RandomState(4))
    V = np.stack([pf_oracle.nr_polar(net, P[i], Q[i])[0] for i in range(120)])
    ref = ss.fit_from_data(ss.interleave(P, Q), V)
    assert np.abs(vp.coef_ - ref.coef_).max() < 1e-6 and np.abs(vp.intercept_ - ref.intercept_).max() < 1e-6


@pytest.mark.parametrize("low,high", [(0.0, 1.0), (-1.0, 1.0)])
def test_env_action_from_the_same_launch_is_translate_action_bit_for_bit(net, series_small, low, high):
    """include/flexenv.h: flexenv_safety_project_env — the adjusted vector and what env.step is fed next (the fp32 cast and
    utils/util.py:125-128's clamp / shift / scale) from one launch: equal, bit for bit, to the cast and the five tensor
    operations on the adjusted vector; the adjusted vector itself is the plain call's; no flags when none are asked for."""
    import types
    import torch
    from safe_marl_amd.flex_env import VecFlexProvisionEnv
    from safe_marl_amd import safety_signal as ss
    from safe_marl_amd.util import scale_action
    vp = ss.fit_voltage_predictor(net, num_scenarios=300)
    sp, sq, beta = vp.building_terms(net)
    n = 1000
    g = torch.Generator(device="cuda").manual_seed(3)
    vec = VecFlexProvisionEnv({"alg": "safemaddpg"}, n, series=series_small, net=net, seed=5)
    vec.reset()
    proposed = torch.rand(n, 5, 4, device="cuda", generator=g) * 2.4 - 1.2
    adj0, hit0 = vec.safety_project(proposed, sp, sq, beta, 0.97, 1.03)
    adj1, hit1, env1 = vec.safety_project(proposed, sp, sq, beta, 0.97, 1.03, env_action_range=(low, high))
    adj2, hit2, env2 = vec.safety_project(proposed, sp, sq, beta, 0.97, 1.03, env_action_range=(low, high), want_hit=False)
    want = scale_action(types.SimpleNamespace(action_low=low, action_high=high), adj0.to(torch.float32))
    assert torch.equal(adj0, adj1) and torch.equal(adj0, adj2) and torch.equal(hit0, hit1) and hit2 is None
    assert env1.dtype == torch.float32 and torch.equal(env1, want) and torch.equal(env2, want)
    assert bool(hit0.any())                                  # the limits bite somewhere: the projection is exercised
