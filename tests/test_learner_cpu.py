"""Learner parity against golden vectors captured by importing the reference's own torch modules
(tests/golden/make_learner_golden.py).  fp32 arithmetic: tolerances are stated per check."""
import json
import os

import numpy as np
import pytest
import torch as th

from safe_marl_amd.util import convert, select_action, translate_action
from safe_marl_amd.learner import MADDPG
from safe_marl_amd.replay_buffer import DeviceReplayBuffer, Transition, TransReplayBuffer
from safe_marl_amd.trainer import PGTrainer

G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def gold():
    return dict(np.load(os.path.join(G, "learner_golden.npz")))


@pytest.fixture(scope="module")
def args():
    return convert(json.load(open(os.path.join(G, "learner_args.json"))))


def _load_sd(name):
    z = np.load(os.path.join(G, name))
    return {k: th.from_numpy(z[k]) for k in z.files}


def _batch(device="cpu"):
    z = np.load(os.path.join(G, "learner_batch.npz"))
    return Transition(**{k: th.from_numpy(z[k]).float().to(device) for k in Transition._fields})


def _model(args):
    target = MADDPG(args)
    model = MADDPG(args, target)
    sd = _load_sd("learner_state_dict.npz")
    missing = model.load_state_dict(sd, strict=True)      # same keys and shapes as the reference's state_dict
    assert not missing.missing_keys and not missing.unexpected_keys
    return model


class StubEnv:
    n_envs = 1

    def get_num_of_agents(self):
        return 5


def test_select_and_translate_action(gold, args):
    means = th.from_numpy(gold["sa_means"])
    log_std = th.zeros_like(means)
    th.manual_seed(11)
    act, logp = select_action(args, means, status="train", exploration=True, info={"log_std": log_std})
    assert np.allclose(act.numpy(), gold["sa_train_explore_action"], atol=1e-7)
    assert np.allclose(logp.numpy(), gold["sa_train_explore_logp"], atol=1e-5)
    act, lp = select_action(args, means, status="train", exploration=False, info={"log_std": log_std})
    assert lp is None and np.array_equal(act.numpy(), gold["sa_train_noexplore_action"])
    act, _ = select_action(args, means, status="test", exploration=False, info={"log_std": log_std})
    assert np.allclose(act.numpy(), gold["sa_test_action"], atol=1e-7)
    raw, env_a = translate_action(args, th.from_numpy(gold["ta_in"]), None)
    assert np.array_equal(raw.numpy(), gold["ta_raw"]) and np.allclose(env_a, gold["ta_env"], atol=1e-7)
    assert env_a.min() >= 0.5 and env_a.max() <= 1.0


def test_policy_value_loss_and_grads(gold, args):
    model = _model(args)
    batch = _batch()
    unpacked = model.unpack_data(batch)
    assert np.allclose(unpacked[5].detach().numpy(), gold["unpack_reward_bn"], atol=2e-5)      # reward BatchNorm
    means, _, hiddens = model.policy(batch.state, last_hid=batch.last_hid)
    assert np.allclose(means.detach().numpy(), gold["policy_means"], atol=2e-6)
    assert np.allclose(hiddens.detach().numpy(), gold["policy_hiddens"], atol=2e-6)
    v = model.value(batch.state, batch.action)
    assert np.allclose(v.detach().numpy(), gold["value_sa"], atol=1e-5)
    model = _model(args)
    policy_loss, value_loss, _ = model.get_loss(batch)
    assert abs(policy_loss.item() - gold["policy_loss"]) < 2e-6
    assert abs(value_loss.item() - gold["value_loss"]) < 1e-5 * max(1.0, abs(gold["value_loss"]))
    model.zero_grad()
    value_loss.backward()
    for k, p in model.value_dicts.named_parameters():
        ref = gold["vgrad." + k]
        assert np.allclose(p.grad.numpy(), ref, atol=2e-6 + 1e-4 * np.abs(ref).max()), k
    policy_loss2, _, _ = model.get_loss(batch)
    model.zero_grad()
    policy_loss2.backward()
    for k, p in model.policy_dicts.named_parameters():
        ref = gold["pgrad." + k]
        assert np.allclose(p.grad.numpy(), ref, atol=2e-7 + 1e-4 * np.abs(ref).max()), k


def test_one_optimizer_step_and_target_update(gold, args):
    trainer = PGTrainer(args, MADDPG, StubEnv(), None)
    sd = _load_sd("learner_state_dict.npz")
    trainer.behaviour_net.load_state_dict(sd)
    batch = _batch()
    stat = {}
    trainer.value_transition_process(stat, batch)
    trainer.policy_transition_process(stat, batch)
    for k in ("mean_train_value_grad_norm", "mean_train_value_loss", "mean_train_policy_grad_norm",
              "mean_train_policy_loss", "mean_train_entropy"):
        assert abs(float(stat[k]) - gold["stat." + k]) < 1e-4 * max(1.0, abs(gold["stat." + k])), k
    after = _load_sd("learner_state_dict_after_step.npz")
    mine = trainer.behaviour_net.state_dict()
    for k, ref in after.items():
        if ref.is_floating_point():
            assert th.allclose(mine[k], ref, atol=3e-6, rtol=1e-5), k
    trainer.behaviour_net.update_target()
    tgt = _load_sd("learner_target_after_update.npz")
    mine_t = trainer.behaviour_net.target_net.state_dict()
    for k, ref in tgt.items():
        if ref.is_floating_point():
            assert th.allclose(mine_t[k], ref, atol=3e-6, rtol=1e-5), k


def test_replay_buffer_window_sequence(gold):
    """replay_buffer.py:3-30: FIFO overflow and contiguous-window sampling from the global NumPy RNG."""
    buf = DeviceReplayBuffer(50)
    ids = []
    np.random.seed(3)
    for i in range(77):
        z = th.zeros(1, 2, 3)
        buf.add_batch(state=z, action=z, log_prob_a=z, value=z, next_value=z, reward=th.full((1, 2), float(i)),
                      next_state=z, done=th.zeros(1), last_step=th.zeros(1), action_avail=z, last_hid=z, hid=z)
        if i % 7 == 6 and len(buf.buffer) >= 8:
            ids.append(buf.get_batch_tensors(8).reward[:, 0].numpy().astype(int).tolist())
    assert np.array_equal(np.array(ids), gold["replay_ids"])
    assert len(buf.buffer) == int(gold["replay_len"]) == 50
    assert [int(t.reward[0]) for t in buf.get_batch(8)] in [list(range(s, s + 8)) for s in range(27, 70)]
    buf.clear()
    assert len(buf.buffer) == 0


def test_add_experience_accepts_reference_transitions(args):
    """The numpy Transition of model.py:230-242 goes in, tensors come out with the unpack_data shapes."""
    buf = TransReplayBuffer(10, device="cpu")
    rng = np.random.default_rng(0)
    for t in range(12):
        buf.add_experience(Transition([rng.normal(size=144) for _ in range(5)], rng.normal(size=(1, 5, 4)).astype(np.float32),
                                      rng.normal(size=(1, 5, 4)), rng.normal(size=(1, 5, 1)), rng.normal(size=(1, 5, 1)),
                                      np.array([0.1 * t] * 5), [rng.normal(size=144) for _ in range(5)], t == 11, t == 11,
                                      np.ones((1, 5, 4)), rng.normal(size=(1, 5, 64)), rng.normal(size=(1, 5, 64))))
    assert len(buf.buffer) == 10
    w = buf.window(0, 10)
    assert w.state.shape == (10, 5, 144) and w.action.shape == (10, 5, 4) and w.last_hid.shape == (10, 5, 64)
    assert np.allclose(w.reward[:, 0].numpy(), 0.1 * np.arange(2, 12), atol=1e-6) and w.done[-1] == 1


def test_transition_update_schedule(gold, args):
    """model.py:40-71: 10 value + 1 policy sub-updates whenever steps % 60 == 0 (and steps > 0, buffer >= 32);
    target update whenever steps % 120 == 0 — including steps == 0, before the first increment."""
    model = _model(args)

    class StubTrainer:
        def __init__(self):
            self.replay_buffer = DeviceReplayBuffer(5000)
            self.steps = 0
            self.log = []

        def effective_batch_size(self):
            return args.batch_size

        def value_replay_process(self, stat):
            self.log.append((self.steps, 0))

        def policy_replay_process(self, stat):
            self.log.append((self.steps, 1))

    st = StubTrainer()
    hits = []
    model.update_target = lambda: hits.append(st.steps)
    z = th.zeros(1, 1)
    for i in range(300):
        st.replay_buffer.add_batch(**{k: z for k in Transition._fields})
        model.transition_update(st, None, {})
        st.steps += 1
    assert np.array_equal(np.array(st.log), gold["sched_calls"])
    assert np.array_equal(np.array(hits), gold["sched_target"])


def test_critic_block_form_equals_explicit_input(args):
    """The critic's column-block evaluation equals fc1 applied to the explicit [obs_all | onehot | acts] rows of
    maddpg.py:33-76, values and own-action gradients alike."""
    model = _model(args)
    th.manual_seed(0)
    b, n = 6, 5
    obs = th.randn(b, n, 144)
    act = th.randn(b, n, 4, requires_grad=True)
    v = model.value(obs, act)
    ids = th.eye(n).unsqueeze(0).expand(b, -1, -1)
    rows = th.cat([obs.reshape(b, 1, -1).expand(b, n, -1), ids, act.reshape(b, 1, -1).expand(b, n, -1)], dim=-1)
    v_ref, _ = model.value_dicts[0](rows.reshape(b * n, -1), None)
    assert th.allclose(v.view(-1), v_ref.view(-1), atol=1e-5)
    g, = th.autograd.grad(v.sum(), act)
    # explicit form with other agents' actions detached
    act2 = act.detach().clone().requires_grad_(True)
    a_rep = act2.unsqueeze(1).expand(b, n, n, 4)
    mask = th.eye(n).view(1, n, n, 1)
    a_in = (a_rep * mask + (a_rep * (1 - mask)).detach()).reshape(b, n, -1)
    rows2 = th.cat([obs.reshape(b, 1, -1).expand(b, n, -1), ids, a_in], dim=-1)
    v2, _ = model.value_dicts[0](rows2.reshape(b * n, -1), None)
    g2, = th.autograd.grad(v2.sum(), act2)
    assert th.allclose(g, g2, atol=1e-5)


def test_matd3_matches_reference(gold, args):
    """madrl/models/matd3.py: twin-flag critic, min-of-twins target, agent-summed action quirk."""
    from safe_marl_amd.learner import MATD3
    target = MATD3(args)
    model = MATD3(args, target)
    res = model.load_state_dict(_load_sd("matd3_state_dict.npz"), strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    assert model.value_dicts[0].fc1.in_features == 746
    batch = _batch()
    v = model.value(batch.state, batch.action)
    assert v.shape == (64, 5, 1) and np.allclose(v.detach().numpy(), gold["matd3_value"], atol=1e-5)
    th.manual_seed(99)
    pl, vl, _ = model.get_loss(batch)
    assert abs(pl.item() - gold["matd3_policy_loss"]) < 2e-6
    assert abs(vl.item() - gold["matd3_value_loss"]) < 1e-5 * max(1.0, abs(gold["matd3_value_loss"]))
    model.zero_grad()
    vl.backward()
    for k, p in model.value_dicts.named_parameters():
        ref = gold["matd3_vgrad." + k]
        assert np.allclose(p.grad.numpy(), ref, atol=2e-6 + 1e-4 * np.abs(ref).max()), k
    th.manual_seed(99)
    pl2, _, _ = model.get_loss(batch)
    model.zero_grad()
    pl2.backward()
    for k, p in model.policy_dicts.named_parameters():
        ref = gold["matd3_pgrad." + k]
        assert np.allclose(p.grad.numpy(), ref, atol=2e-7 + 1e-4 * np.abs(ref).max()), k
    # the single-loss evaluations the trainer uses give the same numbers
    th.manual_seed(99)
    _, vl_only, _ = model.get_loss(batch, need="value")
    pl_only, _, _ = model.get_loss(batch, need="policy")
    assert abs(vl_only.item() - vl.item()) < 1e-6 and abs(pl_only.item() - pl.item()) < 1e-7


def test_iddpg_matches_reference(gold, args):
    """madrl/models/iddpg.py + learning_algorithms/ddpg.py: independent critics on (o_i, id_i, a_i)."""
    from safe_marl_amd.learner import IDDPG
    model = IDDPG(args, IDDPG(args))
    res = model.load_state_dict(_load_sd("iddpg_state_dict.npz"), strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    assert model.value_dicts[0].fc1.in_features == 144 + 4 + 5
    batch = _batch()
    v = model.value(batch.state, batch.action)
    assert np.allclose(v.detach().numpy(), gold["iddpg_value"], atol=1e-5)
    pl, vl, _ = model.get_loss(batch)
    assert abs(pl.item() - gold["iddpg_policy_loss"]) < 2e-6
    assert abs(vl.item() - gold["iddpg_value_loss"]) < 1e-5 * max(1.0, abs(gold["iddpg_value_loss"]))
    model.zero_grad()
    vl.backward()
    for k, p in model.value_dicts.named_parameters():
        ref = gold["iddpg_vgrad." + k]
        assert np.allclose(p.grad.numpy(), ref, atol=2e-6 + 1e-4 * np.abs(ref).max()), k
    pl2, _, _ = model.get_loss(batch)
    model.zero_grad()
    pl2.backward()
    for k, p in model.policy_dicts.named_parameters():
        ref = gold["iddpg_pgrad." + k]
        assert np.allclose(p.grad.numpy(), ref, atol=2e-7 + 1e-4 * np.abs(ref).max()), k


def test_run_lengths_cover_the_training_loops_schedule():
    """Model._run_lengths (which burst graphs the rollout records up front) against the loop of _train_process_graph itself,
    simulated for 500 episodes from several starting counters."""
    import types
    from safe_marl_amd.learner import Model
    for freqs, horizon in (((60, 120), 95), ((60, 120), 96), ((7, 11), 95), ((30, 0), 10)):
        me = types.SimpleNamespace(args=types.SimpleNamespace(behaviour_update_freq=freqs[0], target_update_freq=freqs[1],
                                                              target=freqs[1] > 0))
        fs = [f for f in freqs if f > 0]
        for start in (0, 17, 190):
            want, s = set(), start
            for _ in range(500):
                t = 0
                while t < horizon:
                    m = min([horizon - t] + [(-s) % f + 1 for f in fs])
                    want.add(m)
                    t += m
                    s += m
            assert Model._run_lengths(me, start, horizon) == sorted(want)


def test_three_agent_maddpg_matches_the_reference():
    """BASELINE.json config 3 (3 agents: critic input (144 + 4) * 3 + 3, maddpg.py:18-27): the learner3_* fixtures
    (make_learner_golden.py --agents 3) — values, losses, every gradient, one value + one policy step, target update."""
    gold = dict(np.load(os.path.join(G, "learner3_golden.npz")))
    args = convert(json.load(open(os.path.join(G, "learner3_args.json"))))
    z = np.load(os.path.join(G, "learner3_batch.npz"))
    batch = Transition(**{k: th.from_numpy(z[k]).float() for k in Transition._fields})
    model = MADDPG(args, MADDPG(args))
    res = model.load_state_dict(_load_sd("learner3_state_dict.npz"), strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    assert tuple(model.value_dicts[0].fc1.weight.shape) == (64, 447)
    means, _, hiddens = model.policy(batch.state, last_hid=batch.last_hid)
    assert np.allclose(means.detach().numpy(), gold["policy_means"], atol=2e-6)
    assert np.allclose(model.value(batch.state, batch.action).detach().numpy(), gold["value_sa"], atol=1e-5)
    policy_loss, value_loss, _ = model.get_loss(batch)
    assert abs(policy_loss.item() - gold["policy_loss"]) < 2e-6
    assert abs(value_loss.item() - gold["value_loss"]) < 1e-5 * max(1.0, abs(gold["value_loss"]))
    model.zero_grad()
    value_loss.backward()
    for k, p in model.value_dicts.named_parameters():
        ref = gold["vgrad." + k]
        assert np.allclose(p.grad.numpy(), ref, atol=2e-6 + 1e-4 * np.abs(ref).max()), k

    class Env3:
        n_envs = 1

        def get_num_of_agents(self):
            return 3

    trainer = PGTrainer(args, MADDPG, Env3(), None)
    trainer.behaviour_net.load_state_dict(_load_sd("learner3_state_dict.npz"))
    stat = {}
    trainer.value_transition_process(stat, batch)
    trainer.policy_transition_process(stat, batch)
    after = _load_sd("learner3_state_dict_after_step.npz")
    mine = trainer.behaviour_net.state_dict()
    for k, ref in after.items():
        if ref.is_floating_point():
            assert th.allclose(mine[k], ref, atol=3e-6, rtol=1e-5), k
    trainer.behaviour_net.update_target()
    tgt = _load_sd("learner3_target_after_update.npz")
    mine_t = trainer.behaviour_net.target_net.state_dict()
    for k, ref in tgt.items():
        if ref.is_floating_point():
            assert th.allclose(mine_t[k], ref, atol=3e-6, rtol=1e-5), k
