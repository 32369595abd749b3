"""The two-launch rollout step (fused actor + exploration epilogue, flexnet_rollout_pack) against the PyTorch glue it
replaces (learner.RolloutGraph.body's general path, which follows model.py:198-267 on the batched env)."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(n_envs):
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
    from train_maddpg import DEFAULT_ALG_ARGS
    from safe_marl_amd.flex_env import VecFlexProvisionEnv
    from safe_marl_amd.learner import MADDPG, RolloutGraph
    from safe_marl_amd.network import create_network
    from safe_marl_amd.replay_buffer import TransReplayBuffer
    from safe_marl_amd.series import make_synthetic_series
    from safe_marl_amd.util import convert
    net = create_network()
    series = make_synthetic_series(net, n_days=30)
    alg = dict(DEFAULT_ALG_ARGS)
    alg.update(alg="maddpg", agent_num=5, obs_size=144, state_size=110, action_dim=4)
    out = []
    for _ in range(2):
        torch.manual_seed(11)
        m = MADDPG(convert(alg)).cuda()
        with torch.no_grad():
            for p in m.policy_dicts.parameters():
                p.mul_(10.0)                       # the default init is tiny: make the policy matter
        env = VecFlexProvisionEnv({}, n_envs, net=net, series=series, seed=5, warm_start=True)
        rg = RolloutGraph(m, env, TransReplayBuffer(n_envs * 8, device="cuda"))
        out.append(rg)
    return out


@pytest.mark.parametrize("n_envs", [64, 33])
def test_fast_step_equals_general_step(n_envs):
    fast, slow = _setup(n_envs)
    assert fast.fast
    slow.fast = False
    slow.model.fused_inference = False             # general path end to end: the module, torch glue
    for rg in (fast, slow):
        rg.start_episode(rg.env.reset())
    for step in range(4):
        for rg in (fast, slow):
            torch.manual_seed(100 + step)          # the same exploration draws
            rg.body()
        torch.cuda.synchronize()
        for name in ("rec", "obs", "hid"):
            a, b = getattr(fast, name), getattr(slow, name)
            # two fp32 summation orders through a recurrent net, four steps: relative 5e-4
            assert (a - b).abs().max().item() < 5e-4 * max(1.0, b.abs().max().item()), (step, name, (a - b).abs().max().item())
        assert abs(fast.rew_sum.item() - slow.rew_sum.item()) < 1e-4 * max(1.0, abs(slow.rew_sum.item()))
        assert torch.allclose(fast.info_sum, slow.info_sum, rtol=1e-5, atol=1e-5)
        assert fast.fail_sum.item() == slow.fail_sum.item()
    # the record really is the transition: columns against their sources
    f = fast.f
    assert torch.equal(f["next_state"], fast.env.obs)
    assert torch.equal(f["done"], fast.env.done.float())
    assert torch.allclose(f["reward"], fast.env.reward.float().unsqueeze(1).expand(-1, 5))


def test_graph_capture_with_the_fused_step():
    fast, _ = _setup(128)
    fast.capture()
    fast.start_episode(fast.env.reset())
    before = fast.obs.clone()
    for _ in range(3):
        fast.graph.replay()
    torch.cuda.synchronize()
    assert not torch.equal(before, fast.obs) and torch.isfinite(fast.rec).all()
    assert torch.equal(fast.obs, fast.env.obs)
