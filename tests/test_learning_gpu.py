"""GPU: the vectorised trainer LEARNS.  Short versions of tools/learning_curve.py (whose 400-episode
record is profiles/r03_learning_curve*.json): train for a few dozen episodes and evaluate in test mode (tanh(mean),
util.py:79-82; model.py:269-306) on FIXED evaluation episodes (injected reset draws), before and after.

Margins, from the recorded curves (stand-in IEEE-33 feeder, generated series): MADDPG's test reward per env-step goes
+0.0226 -> +0.0417 within 20 episodes (+0.0415..0.0421 at 400; the OPF comparator reaches +0.0435 on the same episodes);
the asserted margin is +0.012.  On the stressed feeder (all loads x 1.5: under-voltages in ~54 % of env-steps) MADDPG goes
-0.019 -> +0.022 and SAFEMADDPG's violation rate (0.20) lies below MADDPG's (0.51)."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def _args(env, alg="maddpg"):
    from safe_marl_amd.util import convert
    d = json.load(open(os.path.join(G, "learner_args.json")))
    d.update(cuda=True, alg=alg, agent_num=env.n_agents, obs_size=env.obs_size, state_size=env.state_size, action_dim=4,
             v_min=0.9, v_max=1.1)
    return convert(d)


def _spec(series, n, n_agents=5, seed=2025):
    rng = np.random.default_rng(seed)
    return dict(day=rng.integers(0, series.n_start_days(96), n).astype(np.int32), hour=rng.integers(0, 24, n).astype(np.int32),
                interval=rng.integers(0, 4, n).astype(np.int32), e0=rng.uniform(0.01125, 0.01375, (n, n_agents)),
                a0=rng.uniform(0.0, 1.0, (n, 4 * n_agents)))


def _series(net, scale=1.0):
    from safe_marl_amd.series import make_synthetic_series
    s = make_synthetic_series(net, n_days=120)
    if scale != 1.0:
        s.table[:, :2 * len(net["bus_numbers"])] *= scale
    return s


def _train_and_evaluate(net, series, alg, envs, episodes, batch_div=4, n_eval=256, intended=False):
    """(evaluation before, evaluation after, trainer) on the same fixed episodes."""
    import torch
    from safe_marl_amd import learner
    from safe_marl_amd.flex_env import VecFlexProvisionEnv
    from safe_marl_amd.trainer import PGTrainer
    env_args = {"alg": "safemaddpg"} if alg == "safemaddpg" else {}
    env = VecFlexProvisionEnv(env_args, envs, net=net, series=series, seed=5, warm_start=True)
    ev = VecFlexProvisionEnv(env_args, n_eval, net=net, series=series, seed=6, warm_start=True)
    torch.manual_seed(0)
    np.random.seed(0)
    cls = {"maddpg": learner.MADDPG, "safemaddpg": learner.SAFEMADDPG}[alg]
    tr = PGTrainer(_args(env, alg), cls, env, None, batch_scale=max(1, envs // batch_div), replay_capacity=envs * 96 * 2)
    if intended:
        tr.behaviour_net.intended_actions = True
    spec = _spec(series, n_eval)
    before = tr.behaviour_net.evaluate_on(ev, spec)
    again = tr.behaviour_net.evaluate_on(ev, spec)
    assert before == again                                   # fixed episodes, test mode: the evaluation is deterministic
    for _ in range(episodes):
        stat = {}
        tr.behaviour_net.train_process(stat, tr)
    assert sorted(tr._update_graphs) == ["policy", "value"]   # the product path: graphed rollout and sub-updates
    assert tr.behaviour_net._rollout_graph.graph is not None
    after = tr.behaviour_net.evaluate_on(ev, spec)
    for st in (before, after):
        assert all(np.isfinite(v) for v in st.values()) and st["mean_test_solver_failed"] == 0.0
    return before, after, tr


def test_maddpg_learns_at_the_default_batch(net):
    """batch_scale = n_envs / 4: 1.47 samples consumed per transition collected."""
    before, after, tr = _train_and_evaluate(net, _series(net), "maddpg", 1024, 40)
    assert tr.effective_batch_size() == 32 * 256
    assert after["mean_test_reward"] > before["mean_test_reward"] + 0.012, (before, after)
    # where the improvement comes from on this data: the load reduction is worth more than its discomfort
    assert after["mean_test_revenue"] - after["mean_test_discomfort_penalty"] > \
        before["mean_test_revenue"] - before["mean_test_discomfort_penalty"] + 0.012


def test_maddpg_learns_at_the_reference_sample_reuse(net):
    """batch_scale = n_envs: 11 x 32 samples per 60 transitions = 5.87 (model.py:43-50 x replay_buffer.py:17-21)."""
    before, after, tr = _train_and_evaluate(net, _series(net), "maddpg", 512, 40, batch_div=1)
    assert tr.effective_batch_size() == 32 * 512
    assert after["mean_test_reward"] > before["mean_test_reward"] + 0.012, (before, after)


def test_safemaddpg_on_a_stressed_feeder(net):
    """All loads x 1.5: under-voltages in about half of the env-steps.  (1) MADDPG learns there too and lowers the voltage
    penalty; (2) SAFEMADDPG's violation rate is not above MADDPG's; (3) with the reference's action routing (SURVEY A13:
    the safety layer's physical-unit vector goes through translate_action and is re-read raw, so every entry arrives
    >= 0.5 and saturates) the policy cannot move the environment: the test reward is the same before and after training;
    (4) with ``intended_actions`` (NOT the reference's behaviour) it learns."""
    s = _series(net, 1.5)
    b_m, a_m, _ = _train_and_evaluate(net, s, "maddpg", 1024, 40)
    assert b_m["mean_test_violation_rate"] > 0.3                      # the scenario does violate
    assert a_m["mean_test_reward"] > b_m["mean_test_reward"] + 0.02, (b_m, a_m)
    assert a_m["mean_test_voltage_penalty"] < b_m["mean_test_voltage_penalty"]
    b_s, a_s, _ = _train_and_evaluate(net, s, "safemaddpg", 1024, 40)
    assert a_s["mean_test_violation_rate"] <= a_m["mean_test_violation_rate"], (a_s, a_m)
    assert abs(a_s["mean_test_reward"] - b_s["mean_test_reward"]) < 1e-3, (b_s, a_s)
    b_i, a_i, _ = _train_and_evaluate(net, s, "safemaddpg", 1024, 40, intended=True)
    assert a_i["mean_test_reward"] > b_i["mean_test_reward"] + 0.03, (b_i, a_i)
    assert a_i["mean_test_violation_rate"] <= a_m["mean_test_violation_rate"], (a_i, a_m)
