"""GPU, ONE hop: the reference's own select_action / translate_action vectors (utils/util.py:50-85,121-130, captured by
tests/golden/make_learner_golden.py from the imported reference) against the FUSED epilogues of the product path —

* ``flexnet_agent_sum_explore`` (csrc/rollout.hip): the action selection launch of the MATD3 / IDDPG rollout and of both
  losses' policy evaluations; with one agent per row the agent sum is the mean itself;
* the exploration epilogue of the fused actor kernels (csrc/actor.hip, csrc/actor_r16.h): tanh(mean + std * eps) and the
  environment's action 0.5 (clamp(a, low, high) + 1) (high - low) + low, in the launch that evaluates the policy.

The actor kernels take observations, not means, so the golden means are driven THROUGH the network: a policy whose GRU
update gate is saturated (b_hz = 100: z = 1 exactly), whose candidate is tanh(0) = 0 and whose fc2 picks hidden units
0..3 hands the first four entries of the incoming hidden state through as the means, exactly (every other product is a
multiplication by an exact zero) — asserted below before anything is compared with the golden vectors.

Tolerances: the raw mean (train, no exploration) and translate_action are exact fp32 arithmetic -> bit-equal;
tanh: libm's tanhf in flexnet_agent_sum_explore (1e-7), v_exp_f32 / v_rcp_f32 in the actor epilogue (~1 ulp of
the intermediate, 4e-7 absolute here)."""
import os
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def gold():
    return dict(np.load(os.path.join(G, "learner_golden.npz")))


def _agent_sum(means, eps=None, low=0.0, high=1.0):
    """one launch of flexnet_agent_sum_explore on [rows, 1, act] (n_agents = 1: sum over agents = the mean)"""
    import ctypes as C
    from safe_marl_amd import _lib
    rows, act = means.shape
    means = means.contiguous()
    action, env_action = torch.empty_like(means), torch.empty_like(means)
    std = torch.ones(act, device="cuda")
    k = _lib.FlexAgentSumArgs()
    k.n_envs, k.n_agents, k.act_dim = rows, 1, act
    k.act_low, k.act_high = low, high
    k.means, k.action, k.env_action = means.data_ptr(), action.data_ptr(), env_action.data_ptr()
    if eps is not None:
        k.eps, k.std = eps.data_ptr(), std.data_ptr()
    _lib.check(_lib.load().flexnet_agent_sum_explore(C.byref(k), C.c_void_p(torch.cuda.current_stream().cuda_stream)),
               "flexnet_agent_sum_explore")
    torch.cuda.synchronize()
    return action.cpu().numpy(), env_action.cpu().numpy()


def test_agent_sum_explore_launch_reproduces_select_and_translate_action(gold):
    means = torch.from_numpy(gold["sa_means"]).cuda().reshape(15, 4)
    zeros = torch.zeros(15, 4, device="cuda")
    # train, no exploration (util.py:75-77): the mean itself
    act, _ = _agent_sum(means)
    assert np.array_equal(act.reshape(3, 5, 4), gold["sa_train_noexplore_action"])
    # test mode under action_enforcebound (util.py:79-82) = the exploration formula with a zero draw: tanh(mean)
    act, env = _agent_sum(means, zeros)
    assert np.abs(act.reshape(3, 5, 4) - gold["sa_test_action"]).max() <= 1e-7
    # translate_action (util.py:125-128) of the golden input: raw action handed through, env action scaled
    raw, env = _agent_sum(torch.from_numpy(gold["ta_in"]).cuda().reshape(5, 4))
    assert np.array_equal(raw.reshape(gold["ta_raw"].shape), gold["ta_raw"])
    assert np.array_equal(env.reshape(gold["ta_env"].shape), gold["ta_env"].astype(np.float32))
    assert env.min() >= 0.5 and env.max() <= 1.0


def _pass_through_agent(n_agents):
    """RNNAgent (rnn_agent.py:13-33) whose means are hidden_in[:, :4], exactly."""
    from safe_marl_amd.nets import RNNAgent
    args = types.SimpleNamespace(hid_size=64, layernorm=True, action_dim=4, agent_num=n_agents, hid_activation="relu")
    agent = RNNAgent(144 + n_agents, args).cuda()
    with torch.no_grad():
        for p in agent.rnn.parameters():
            p.zero_()
        agent.rnn.bias_hh[64:128] = 100.0          # z = sigmoid(100) = 1: h' = h
        agent.fc2.weight.zero_()
        agent.fc2.bias.zero_()
        for j in range(4):
            agent.fc2.weight[j, j] = 1.0
    return agent


# 0: the library's choice (five 16-row tiles per CU at these sizes), 1: VALU kernel, 3: 32-row matrix-core kernel
@pytest.mark.parametrize("variant", [0, 1, 3])
@pytest.mark.parametrize("tile", [1, 1400])                 # 15 rows; 21 000 rows (past the 16-row kernel's range)
def test_actor_kernel_epilogue_reproduces_select_and_translate_action(gold, variant, tile):
    from safe_marl_amd.nets import fused_actor_forward
    n = 5
    agent = _pass_through_agent(n)
    g = torch.Generator(device="cuda").manual_seed(2)

    def run(m):                                             # m: [b, 5, 4] golden means -> (means, action, env action)
        b = m.shape[0]
        obs = torch.randn(b * tile, n, 144, device="cuda", generator=g)
        hid = torch.randn(b * tile, n, 64, device="cuda", generator=g)
        hid[:, :, :4] = torch.from_numpy(m).cuda().repeat(tile, 1, 1)
        noise = torch.zeros(b * tile, n, 4, device="cuda")
        out = fused_actor_forward(agent, obs, hid, n, True, noise=noise, std=1.0, low=0.0, high=1.0, variant=variant)
        assert out is not None
        torch.cuda.synchronize()
        means, hid_out, action, env_action = out
        assert torch.equal(hid_out.reshape(b * tile, n, 64), hid)                      # the pass-through really is exact
        for t in (means, action, env_action):                                          # every tile holds the same rows
            assert torch.equal(t.reshape(tile, b * n * 4)[0], t.reshape(tile, b * n * 4)[-1])
        return [t.reshape(b * tile, n, 4)[:b].cpu().numpy() for t in (means, action, env_action)]

    means, action, env_action = run(gold["sa_means"])
    assert np.array_equal(means, gold["sa_train_noexplore_action"])                    # util.py:75-77
    assert np.abs(action - gold["sa_test_action"]).max() <= 4e-7                       # util.py:79-82 (zero draw)
    means, action, env_action = run(gold["ta_x"])
    assert np.abs(action.reshape(gold["ta_raw"].shape) - gold["ta_raw"]).max() <= 4e-7
    assert np.abs(env_action.reshape(gold["ta_env"].shape) - gold["ta_env"]).max() <= 4e-7          # util.py:125-128
    # and the scaling itself is exact on the kernel's own action
    assert np.array_equal(env_action, (0.5 * (np.clip(action, 0.0, 1.0) + np.float32(1.0))).astype(np.float32))
