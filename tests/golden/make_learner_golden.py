#!/usr/bin/env python3
"""Generates tests/golden/learner_*.npz by IMPORTING the reference's torch/numpy-only modules
(utils/replay_buffer.py, utils/util.py, utils/trainer.py, madrl/models/{model,maddpg}.py,
madrl/agents/rnn_agent.py, madrl/critics/mlp_critic.py) from /root/reference and recording their
inputs and outputs on seeded synthetic data.  Run in the build container only:

    cd /root/reference && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/make_learner_golden.py
    cd /root/reference && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/make_learner_golden.py --agents 3

The second form writes the learner3_* set: MADDPG with THREE agents (BASELINE.json config 3: critic input
(obs + act) * n + n = 447 + 3, maddpg.py:18-27), sections (2)-(4) only.

The fixtures are data (inputs + expected outputs); no reference source travels.  SURVEY.md §8(c)
lists the seven vector sets captured here.
"""
import json
import os
import sys

import numpy as np
import torch as th
import yaml

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)
os.chdir(REF)

from utils.util import select_action, translate_action, convert  # noqa: E402
from utils.replay_buffer import TransReplayBuffer  # noqa: E402
from utils.trainer import PGTrainer  # noqa: E402
from madrl.models.maddpg import MADDPG  # noqa: E402


N_AGENTS = 5
if "--agents" in sys.argv:
    N_AGENTS = int(sys.argv[sys.argv.index("--agents") + 1])
OUT_DIR = os.environ.get("GOLDEN_OUT", OUT)          # (the judge's reproduction check writes elsewhere)
PREFIX = "learner" if N_AGENTS == 5 else f"learner{N_AGENTS}"


def load_args():
    with open("madrl/args/default.yaml") as f:
        d = yaml.safe_load(f)
    with open("madrl/args/alg_args/maddpg.yaml") as f:
        a = yaml.safe_load(f)["alg_args"]
    with open("madrl/args/env_args/flex_provision.yaml") as f:
        e = yaml.safe_load(f)["env_args"]
    a["action_low"] = e.get("action_low", 0.0)
    a["action_high"] = e.get("action_high", 1.0)
    a["action_bias"] = e.get("action_bias", 0.0)
    a["action_scale"] = e.get("action_scale", 1.0)
    a["alg"] = "maddpg"
    d = {**d, **a}
    d.update(agent_num=N_AGENTS, obs_size=144, state_size=3 * 33 + 2 * N_AGENTS + 1, action_dim=4, cuda=False)
    return d


class StubEnv:
    def get_num_of_agents(self):
        return N_AGENTS


def sd_to_np(sd, prefix=""):
    return {prefix + k: v.detach().cpu().numpy().copy() for k, v in sd.items()}


def synthetic_transitions(model, rng, count):
    """Transitions with the field shapes model.py:230-242 stores."""
    n, o, a, h = N_AGENTS, 144, 4, 64
    out = []
    for t in range(count):
        state = [rng.normal(0, 0.3, o) for _ in range(n)]
        next_state = [rng.normal(0, 0.3, o) for _ in range(n)]
        action = rng.normal(0, 0.5, (1, n, a)).astype(np.float32)
        out.append(model.Transition(
            state, action, rng.normal(0, 1, (1, n, a)).astype(np.float32),
            rng.normal(0, 1, (1, n, 1)).astype(np.float32), rng.normal(0, 1, (1, n, 1)).astype(np.float32),
            np.array([float(rng.normal(0.03, 0.02))] * n), next_state, bool(t % 17 == 16), bool(t % 17 == 16),
            np.ones((1, n, a)), rng.normal(0, 0.2, (1, n, h)).astype(np.float32),
            rng.normal(0, 0.2, (1, n, h)).astype(np.float32)))
    return out


def pack(transitions):
    """Dense arrays of a transition list, field by field (what the build's device buffer stores)."""
    f = lambda k: np.stack([np.asarray(getattr(t, k), dtype=np.float64) for t in transitions])
    return dict(state=f("state"), action=f("action")[:, 0], reward=f("reward"), next_state=f("next_state"),
                done=np.array([t.done for t in transitions], np.float64),
                last_step=np.array([t.last_step for t in transitions], np.float64),
                last_hid=f("last_hid")[:, 0], hid=f("hid")[:, 0], log_prob_a=f("log_prob_a")[:, 0],
                value=f("value")[:, 0], next_value=f("next_value")[:, 0], action_avail=f("action_avail")[:, 0])


def main():
    argd = load_args()
    args = convert(argd)
    json.dump(argd, open(os.path.join(OUT_DIR, PREFIX + "_args.json"), "w"), indent=1, sort_keys=True)

    # (1) select_action / translate_action  (util.py:50-85, 121-130)
    g = {}
    th.manual_seed(7)
    means = th.randn(3, 5, 4) * 1.5
    log_std = th.zeros_like(means)
    g["sa_means"] = means.numpy()
    th.manual_seed(11)
    act, logp = select_action(args, means, status="train", exploration=True, info={"log_std": log_std})
    g["sa_train_explore_action"], g["sa_train_explore_logp"] = act.numpy(), logp.numpy()
    act, _ = select_action(args, means, status="train", exploration=False, info={"log_std": log_std})
    g["sa_train_noexplore_action"] = act.numpy()
    act, _ = select_action(args, means, status="test", exploration=False, info={"log_std": log_std})
    g["sa_test_action"] = act.numpy()
    ta_x = th.randn(1, 5, 4)                # (kept: the GPU test drives the fused tanh epilogue to ta_in from here)
    one = th.tanh(ta_x)
    g["ta_x"], g["ta_in"] = ta_x.numpy(), one.numpy()
    raw, cp = translate_action(args, one, None)
    g["ta_raw"], g["ta_env"] = raw.numpy(), cp

    # (2) MADDPG with a seeded state_dict: policy / value / get_loss + grads  (maddpg.py:33-123)
    th.manual_seed(1234)
    target = MADDPG(args)
    model = MADDPG(args, target)
    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}   # a snapshot, not live views
    tgt0 = {k: v.detach().clone() for k, v in target.state_dict().items()}
    np.savez_compressed(os.path.join(OUT_DIR, PREFIX + "_state_dict.npz"), **sd_to_np(sd0))
    rng = np.random.default_rng(5)
    trans = synthetic_transitions(model, rng, 40)
    batch32 = trans[3:35]
    np.savez_compressed(os.path.join(OUT_DIR, PREFIX + "_batch.npz"), **pack(batch32))
    batch = model.Transition(*zip(*batch32))
    unpacked = model.unpack_data(batch)
    g["unpack_reward_bn"] = unpacked[5].detach().numpy()                       # (6) model.py:321-322
    state_t, last_hid_t = unpacked[0], unpacked[10]
    means, log_stds, hiddens = model.policy(state_t, last_hid=last_hid_t)
    g["policy_means"], g["policy_hiddens"] = means.detach().numpy(), hiddens.detach().numpy()
    g["value_sa"] = model.value(state_t, unpacked[1]).detach().numpy()
    policy_loss, value_loss, _ = model.get_loss(batch)
    g["policy_loss"], g["value_loss"] = policy_loss.item(), value_loss.item()
    model.zero_grad()
    value_loss.backward()
    for k, p in model.value_dicts.named_parameters():
        g["vgrad." + k] = p.grad.numpy().copy()
    policy_loss2, _, _ = model.get_loss(batch)
    model.zero_grad()
    policy_loss2.backward()
    for k, p in model.policy_dicts.named_parameters():
        g["pgrad." + k] = p.grad.numpy().copy()

    # (3) one value_transition_process + one policy_transition_process through PGTrainer (trainer.py:81-108)
    th.manual_seed(1234)
    trainer = PGTrainer(args, MADDPG, StubEnv(), None)
    trainer.behaviour_net.load_state_dict(sd0)
    trainer.behaviour_net.target_net.load_state_dict(tgt0)
    stat = {}
    trainer.value_transition_process(stat, batch)
    trainer.policy_transition_process(stat, batch)
    for k, v in stat.items():
        g["stat." + k] = float(v)
    np.savez_compressed(os.path.join(OUT_DIR, PREFIX + "_state_dict_after_step.npz"), **sd_to_np(trainer.behaviour_net.state_dict()))

    # (4) update_target before/after (model.py:28-38) on the post-step weights
    trainer.behaviour_net.update_target()
    np.savez_compressed(os.path.join(OUT_DIR, PREFIX + "_target_after_update.npz"),
                        **sd_to_np(trainer.behaviour_net.target_net.state_dict()))

    if N_AGENTS != 5:                      # the other sections do not depend on the agent count
        np.savez_compressed(os.path.join(OUT_DIR, PREFIX + "_golden.npz"), **g)
        print("wrote", sorted(f for f in os.listdir(OUT_DIR) if f.startswith(PREFIX + "_")))
        return

    # (5) TransReplayBuffer fill -> overflow -> get_batch index sequence (replay_buffer.py:3-30)
    buf = TransReplayBuffer(50)
    ids = []
    np.random.seed(3)
    for i in range(77):
        buf.add_experience(("t", i))
        if i % 7 == 6 and len(buf.buffer) >= 8:
            ids.append([x[1] for x in buf.get_batch(8)])
    g["replay_ids"] = np.array(ids)
    g["replay_len"] = len(buf.buffer)

    # (7) transition_update firing schedule with a stub trainer (model.py:40-71)
    class StubTrainer:
        def __init__(self):
            self.replay_buffer = TransReplayBuffer(int(args.replay_buffer_size))
            self.steps = 0
            self.log = []

        def value_replay_process(self, stat):
            self.log.append((self.steps, 0))

        def policy_replay_process(self, stat):
            self.log.append((self.steps, 1))

    st = StubTrainer()
    n_target = []
    orig = model.update_target
    model.update_target = lambda: n_target.append(st.steps)
    for i in range(300):
        model.transition_update(st, ("t", i), {})
        st.steps += 1
    model.update_target = orig
    g["sched_calls"] = np.array(st.log)
    g["sched_target"] = np.array(n_target)

    # (8) MATD3 (madrl/models/matd3.py): value(), get_loss and grads with a seeded state_dict
    from madrl.models.matd3 import MATD3
    th.manual_seed(4321)
    t3 = MATD3(args)
    m3 = MATD3(args, t3)
    np.savez_compressed(os.path.join(OUT_DIR, "matd3_state_dict.npz"),
                        **sd_to_np({k: v.detach().clone() for k, v in m3.state_dict().items()}))
    up = m3.unpack_data(batch)
    g["matd3_value"] = m3.value(up[0], up[1]).detach().numpy()
    th.manual_seed(99)
    pl, vl, _ = m3.get_loss(batch)
    g["matd3_policy_loss"], g["matd3_value_loss"] = pl.item(), vl.item()
    m3.zero_grad()
    vl.backward()
    for k, p_ in m3.value_dicts.named_parameters():
        g["matd3_vgrad." + k] = p_.grad.numpy().copy()
    th.manual_seed(99)
    pl2, _, _ = m3.get_loss(batch)
    m3.zero_grad()
    pl2.backward()
    for k, p_ in m3.policy_dicts.named_parameters():
        g["matd3_pgrad." + k] = p_.grad.numpy().copy()

    # (9) IDDPG (madrl/models/iddpg.py + learning_algorithms/ddpg.py): value(), losses and grads
    from madrl.models.iddpg import IDDPG
    th.manual_seed(2468)
    ti = IDDPG(args)
    mi = IDDPG(args, ti)
    np.savez_compressed(os.path.join(OUT_DIR, "iddpg_state_dict.npz"),
                        **sd_to_np({k: v.detach().clone() for k, v in mi.state_dict().items()}))
    up = mi.unpack_data(batch)
    g["iddpg_value"] = mi.value(up[0], up[1]).detach().numpy()
    pl, vl, _ = mi.rl.get_loss(batch, mi, mi.target_net)
    g["iddpg_policy_loss"], g["iddpg_value_loss"] = pl.item(), vl.item()
    mi.zero_grad()
    vl.backward()
    for k, p_ in mi.value_dicts.named_parameters():
        g["iddpg_vgrad." + k] = p_.grad.numpy().copy()
    pl2, _, _ = mi.rl.get_loss(batch, mi, mi.target_net)
    mi.zero_grad()
    pl2.backward()
    for k, p_ in mi.policy_dicts.named_parameters():
        g["iddpg_pgrad." + k] = p_.grad.numpy().copy()

    np.savez_compressed(os.path.join(OUT_DIR, PREFIX + "_golden.npz"), **g)
    print("wrote", sorted(os.listdir(OUT_DIR)))


if __name__ == "__main__":
    main()
