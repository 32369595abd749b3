"""GPU: the HIP environment (through the C ABI) against fixtures produced by RUNNING THE REFERENCE'S OWN ENVIRONMENT CODE
(tests/golden/make_env_golden.py; see tests/test_env_golden_cpu.py for what the three substitutions of that script leave
pinned: everything of SURVEY §8 rows a3-a12 but the numerical solve, whose voltages are the oracle's Newton-Raphson).
(1) the vectorised env on the recorded episodes — injected reset draws, the recorded actions — step by step;
(2) the N = 1 drop-in view constructed and seeded exactly as the reference's env was: same global-NumPy draws, same
    episode, same return types, over a whole episode, a second reset() and a manual_reset().
Tolerances: reward / info / V / E / state <= 1e-10 (north_star: 1e-6); done / steps exact; observations equal after the
fp32 cast their consumer applies (util.py:145)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden", "env_golden.npz")
TOL = 1e-10
INFO = ("reward", "revenue", "der_cost", "ess_cost", "discomfort_penalty", "voltage_penalty", "cumulative_reward")


@pytest.fixture(scope="module")
def gold():
    return dict(np.load(G, allow_pickle=False))


def _series(gold):
    from safe_marl_amd.series import SeriesTable
    a, r, pv, pr = gold["series.active"], gold["series.reactive"], gold["series.pv"], gold["series.price"]
    z = np.zeros((a.shape[0], 1))
    return SeriesTable(np.ascontiguousarray(np.hstack([z, a, z, r, pv, pr])), 33, 5, int(gold["series.time_delta"]))


def _episode(gold, tag, j):
    p = f"{tag}.ep{j}."
    return {k[len(p):]: v for k, v in gold.items() if k.startswith(p)}


@pytest.mark.parametrize("tags,alg,solver,warm", [((("A", 0), ("A", 1), ("B", 0), ("B", 1)), None, 2, True),
                                                  ((("A", 0), ("B", 1), ("A", 1)), None, 0, False),
                                                  ((("C", 0),), "safemaddpg", 2, True)])
def test_vectorised_env_reproduces_the_reference_episodes(net, gold, tags, alg, solver, warm):
    import torch
    from safe_marl_amd.flex_env import VecFlexProvisionEnv
    eps = [_episode(gold, t, j) for t, j in tags]
    n, T = len(eps), min(len(e["reward"]) for e in eps)
    vec = VecFlexProvisionEnv({"alg": alg} if alg else {}, n, series=_series(gold), net=net, solver=solver, warm_start=warm)
    spec = dict(day=np.array([e["start"][0] for e in eps], np.int32), hour=np.array([e["start"][1] for e in eps], np.int32),
                interval=np.array([e["start"][2] for e in eps], np.int32), e0=np.stack([e["e0"] for e in eps]),
                a0=np.stack([e["a0"] for e in eps]))
    obs = vec.reset(spec=spec).cpu().numpy()
    assert vec.failed.sum().item() == 0
    for i, e in enumerate(eps):
        assert np.allclose(obs[i], e["obs"][0].astype(np.float32), rtol=2e-7, atol=0)
    assert np.abs(vec.get_state().cpu().numpy() - np.stack([e["state"][0] for e in eps])).max() < TOL
    assert np.abs(vec.peek("V").cpu().numpy() - np.stack([e["V"][0] for e in eps])).max() < TOL
    assert np.array_equal(vec.peek("E_INIT").cpu().numpy(), np.stack([e["Einit"][0] for e in eps]))        # A5
    for t in range(T):
        acts = np.stack([e["actions"][t] for e in eps]).reshape(n, 5, 4)
        # alternate the two forms of get_obs(): the stacked copy and the row push + view (bit-identical by test_env_gpu.py)
        if t % 2:
            reward, done, info = vec.step(torch.from_numpy(acts).cuda(), fuse_obs=True)
            obs = vec.obs.cpu().numpy()
        else:
            reward, done, info = vec.step(torch.from_numpy(acts).cuda(), obs_rows=True)
            obs = vec.obs_view().cpu().numpy()
        reward, done, info = reward.cpu().numpy(), done.cpu().numpy(), info.cpu().numpy()
        state, v, en = vec.get_state().cpu().numpy(), vec.peek("V").cpu().numpy(), vec.peek("E").cpu().numpy()
        for i, e in enumerate(eps):
            assert abs(reward[i] - e["reward"][t]) < TOL, (t, i)
            assert bool(done[i]) == bool(e["done"][t])
            assert np.abs(info[i] - e["info"][t]).max() < TOL, (t, i)
            assert np.allclose(obs[i], e["obs"][t + 1].astype(np.float32), rtol=2e-7, atol=0), (t, i)
            assert np.abs(state[i] - e["state"][t + 1]).max() < TOL and np.abs(v[i] - e["V"][t + 1]).max() < TOL
            assert np.abs(en[i] - e["E"][t + 1]).max() < TOL
        assert np.array_equal(vec.peek("STEPS").cpu().numpy(), np.array([e["steps"][t + 1] for e in eps]))
    assert vec.failed.sum().item() == 0


@pytest.mark.parametrize("tag", ["A", "B"])
def test_drop_in_env_replays_the_reference_run_from_its_seed(net, gold, tag):
    """FlexibilityProvisionEnv(kwargs) with the reference's seed: np.random.seed (env:49), the draws of reset (env:85-87,
    100,103) in the reference's order from the GLOBAL NumPy stream, then the recorded actions — the reference's own run."""
    from safe_marl_amd.flex_env import FlexibilityProvisionEnv
    seed = int(gold[tag + ".seed"])
    env = FlexibilityProvisionEnv({"seed": seed}, net=net, series=_series(gold), warm_start=True)
    np.random.seed(seed)                     # (make_env_golden.py re-seeds and resets once more after construction)
    first = env.reset()
    n_eps = int(gold[tag + ".episodes"])
    for j in range(n_eps):
        e = _episode(gold, tag, j)
        if j > 0:
            first = env.manual_reset(*[int(x) for x in e["start"]]) if tag == "B" else env.reset()
        obs, state = first
        assert isinstance(obs, list) and len(obs) == 5 and obs[0].shape == (144,)
        assert env.vec.peek("START").item() == int(e["start"][2]) + int(e["start"][1]) * 4 + int(e["start"][0]) * 96
        assert np.allclose(np.stack(obs).astype(np.float32), e["obs"][0].astype(np.float32), rtol=2e-7, atol=0)
        assert np.abs(state - e["state"][0]).max() < TOL
        for t in range(len(e["reward"])):
            r, d, info = env.step(e["actions"][t].astype(np.float32).reshape(5, 4))
            assert isinstance(r, float) and isinstance(d, bool)
            assert abs(r - e["reward"][t]) < TOL and d == bool(e["done"][t])
            assert np.abs(np.array([info[k] for k in INFO]) - e["info"][t]).max() < TOL
            nxt = env.get_obs()
            assert np.allclose(np.stack(nxt).astype(np.float32), e["obs"][t + 1].astype(np.float32), rtol=2e-7, atol=0)
            assert np.abs(env.get_state() - e["state"][t + 1]).max() < TOL
        assert env.steps == 96
    env.close()


def test_whole_training_loop_replays_the_reference_run(net, gold):
    """End to end, against the reference's EXECUTED training loop (tests/golden/make_loop_golden.py: utils/trainer.PGTrainer +
    MADDPG + TransReplayBuffer + the reference env, three episodes of model.py:198-267 = 285 env steps, four update events of
    ten value + one policy sub-update, two soft target updates, then Model.evaluation's ten test-mode episodes
    (model.py:269-306), on CPU from fixed seeds; solve := the oracle's NR):
    the PRODUCT's N = 1 path — the drop-in env on the HIP kernels, the package's trainer / learner / replay on CPU tensors (so
    that torch's CPU generator draws the reference's exploration noise and the global NumPy stream its episode draws and
    replay windows) — from the same initial weights and seeds lands on the same 285 + 950 actions, rewards and dones, the same
    episode and evaluation statistics and the same final weights of behaviour AND target networks."""
    import json
    import torch as th
    from safe_marl_amd.flex_env import FlexibilityProvisionEnv
    from safe_marl_amd.learner import MADDPG
    from safe_marl_amd.trainer import PGTrainer
    from safe_marl_amd.util import convert
    lg = dict(np.load(os.path.join(os.path.dirname(G), "loop_golden.npz"), allow_pickle=False))
    env = FlexibilityProvisionEnv({"seed": 11}, net=net, series=_series(gold), warm_start=True)
    argd = json.loads(str(lg["alg_args_json"]))
    assert argd["cuda"] is False and argd["obs_size"] == env.get_obs_size() and argd["state_size"] == env.get_state_size()
    th.manual_seed(2024)
    trainer = PGTrainer(convert(argd), MADDPG, env, None)
    assert trainer.device.type == "cpu"
    sd = {k[len("init."):]: th.from_numpy(v) for k, v in lg.items() if k.startswith("init.")}
    res = trainer.behaviour_net.load_state_dict(sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    log = {"action": [], "reward": [], "done": []}
    real_step = env.step

    def step(actions):
        r, d, info = real_step(actions)
        log["action"].append(np.array(actions, dtype=np.float64).reshape(-1).copy())
        log["reward"].append(float(r)); log["done"].append(bool(d))
        return r, d, info

    env.step = step
    np.random.seed(11)
    th.manual_seed(7)
    stats = []
    for _ in range(3):
        stat = {}
        trainer.behaviour_net.train_process(stat, trainer)
        stats.append({k: float(v) for k, v in stat.items()})
    assert trainer.steps == int(lg["steps"]) == 285
    ev = {}
    trainer.behaviour_net.evaluation(ev, trainer)                 # model.py:269-306: ten test-mode episodes
    assert len(log["reward"]) == 285 + int(lg["eval_steps"]) == 285 + 950
    for k, ref in zip([str(k) for k in lg["eval_keys"]], lg["eval"]):
        assert abs(float(ev[k]) - float(ref)) <= 2e-6 * max(1.0, abs(float(ref))), (k, float(ev[k]), float(ref))
    # utils/tester.py:16-70 on the trained net: the record test_agent.py pickles, key by key (env:740-778 accessors)
    from safe_marl_amd.tester import PGTester, RECORD_KEYS
    record = PGTester(convert(argd), trainer.behaviour_net, env).run(3, 7, 1)
    assert set(record) == set(RECORD_KEYS) == {k[len("record."):] for k in lg if k.startswith("record.")}
    for k in RECORD_KEYS:
        mine = np.array([np.asarray(x, dtype=np.float64).reshape(-1) for x in record[k]])
        assert mine.shape == lg["record." + k].shape == (96, lg["record." + k].shape[1]), k
        assert np.abs(mine - lg["record." + k]).max() < 1e-6, (k, np.abs(mine - lg["record." + k]).max())
    n_rec = 95
    act, rew, done = np.array(log["action"])[:-n_rec], np.array(log["reward"])[:-n_rec], np.array(log["done"])[:-n_rec]
    assert np.array_equal(done, lg["done"][:len(done)]) and len(done) == 285 + 950
    d_act, d_rew = np.abs(act - lg["action"][:len(act)]).max(), np.abs(rew - lg["reward"][:len(rew)]).max()
    # fp32 policy arithmetic in two summation orders through 285 steps and 44 optimiser steps: measured 1e-6 / 1e-8
    assert d_act < 2e-5 and d_rew < 2e-7, (d_act, d_rew)
    keys = [str(k) for k in lg["stat_keys"]]
    for j, s in enumerate(stats):
        for i, k in enumerate(keys):
            ref = float(lg["stats"][j, i])
            if np.isfinite(ref) and k in s:
                assert abs(s[k] - ref) <= 2e-5 * max(1.0, abs(ref)), (j, k, s[k], ref)
    worst = 0.0
    for k, v in trainer.behaviour_net.state_dict().items():
        ref = th.from_numpy(lg["final." + k])
        if ref.is_floating_point():
            worst = max(worst, float((v.cpu() - ref).abs().max()))
            assert th.allclose(v.cpu(), ref, atol=2e-5, rtol=1e-4), (k, float((v.cpu() - ref).abs().max()))
        else:
            assert th.equal(v.cpu(), ref), k
    print(f"loop replay: |d action| {d_act:.2e}, |d reward| {d_rew:.2e}, |d weight| {worst:.2e}")
    env.close()
