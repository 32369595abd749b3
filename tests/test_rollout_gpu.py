"""The rollout step (fused actor + exploration epilogue, env kernel, flexnet_rollout_pack into the slab replay ring)
against the PyTorch glue it replaces (model.py:198-267 on the batched env, spelled out here with plain tensor ops)."""
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


def _fields(rg):
    t = rg.last_transition()
    return {k: getattr(t, k) for k in ("state", "action", "reward", "next_state", "done", "last_step", "last_hid", "hid")}


@pytest.mark.parametrize("n_envs", [64, 33])
def test_fast_step_equals_general_step(n_envs):
    fast, slow = _setup(n_envs)
    assert fast.fast
    fast.torch_noise = True                        # the same torch.randn draws as the general path
    slow.fast = False
    slow.model.fused_inference = False             # general path end to end: the module, torch glue
    for rg in (fast, slow):
        rg.start_episode(rg.env.reset())
    want_info = torch.zeros_like(slow.info_sum)
    want_rew = torch.zeros_like(slow.rew_sum)
    for step in range(4):
        prev_obs, prev_hid = slow.obs.clone(), slow.hid.clone()
        for rg in (fast, slow):
            torch.manual_seed(100 + step)          # the same exploration draws
            rg.step()
        torch.cuda.synchronize()
        # the pack kernel's statistics blocks (fixed-order fp64 block sums) against plain tensor sums of the env outputs
        want_info += slow.env.info.sum(0)
        want_rew += slow.env.reward.sum()
        assert torch.allclose(slow.info_sum, want_info, rtol=1e-12, atol=1e-12)
        assert torch.allclose(slow.rew_sum, want_rew, rtol=1e-12, atol=1e-12)
        ff, fs = _fields(fast), _fields(slow)
        for name in ff:
            a, b = ff[name], fs[name]
            # two fp32 summation orders through a recurrent net, four steps: relative 5e-4
            assert (a - b).abs().max().item() < 5e-4 * max(1.0, b.abs().max().item()), (step, name, (a - b).abs().max().item())
        assert (fast.hid - slow.hid).abs().max().item() < 5e-4
        assert abs(fast.rew_sum.item() - slow.rew_sum.item()) < 1e-4 * max(1.0, abs(slow.rew_sum.item()))
        assert torch.allclose(fast.info_sum, slow.info_sum, rtol=1e-5, atol=1e-5)
        assert fast.fail_sum.item() == slow.fail_sum.item()
        # the ring really holds the transition of model.py:230-242: every field against its source
        assert torch.equal(fs["state"], prev_obs) and torch.equal(fs["last_hid"], prev_hid)
        assert torch.equal(fs["next_state"], slow.env.obs) and torch.equal(fs["hid"], slow.hid)
        assert torch.equal(fs["done"], slow.env.done.float()) and torch.equal(fs["last_step"], slow.env.done.float())
        assert torch.equal(fs["reward"], slow.env.reward.float().unsqueeze(1).expand(-1, 5))
    # the cursor cells followed the steps: the sink protocol's policy-read cell leads its env-read cell by one slab, the
    # pack protocol's cells agree (include/flexenv.h, include/flexnet.h)
    assert fast.buf.k == 4 and fast.buf.cursor.tolist() == [4, 3] and slow.buf.cursor.tolist() == [4, 4]
    assert fast.sink_active and fast.ring_active and not slow.ring_active     # env-filed transitions against pack + tensors
    assert len(fast.buf.buffer) == 4 * n_envs


def test_graph_capture_with_the_fused_step():
    fast, _ = _setup(128)
    fast.start_episode(fast.env.reset())
    fast.capture()
    fast.start_episode(fast.env.reset())
    before = fast.obs.clone()
    for _ in range(3):
        fast.step()
    torch.cuda.synchronize()
    assert not torch.equal(before, fast.obs) and all(torch.isfinite(v).all() for v in _fields(fast).values())
    assert fast.buf.k == 3 and fast.buf.cursor.tolist() == [3, 2]
    # the exploration noise is drawn anew on every replay (graph-safe Philox offsets), not frozen at capture: with a policy
    # that ignores its inputs (all weights zero: mean = fc2's bias = 0) two replays differ exactly by their draws
    with torch.no_grad():
        for p_ in fast.model.policy_dicts.parameters():
            p_.zero_()
    acts = []
    for _ in range(2):
        fast.step()
        torch.cuda.synchronize()
        acts.append(_fields(fast)["action"].clone())
    assert (acts[0] - acts[1]).abs().mean().item() > 1e-3
    assert not fast.torch_noise and int(fast.rng_state[1].item()) >= 5        # the in-kernel stream: one step per replay


def test_graph_replays_with_torch_noise_too():
    fast, _ = _setup(64)
    fast.torch_noise = True
    fast.start_episode(fast.env.reset())
    fast.capture()
    fast.start_episode(fast.env.reset())
    with torch.no_grad():                                  # (a policy that ignores its inputs: replays differ by their draws)
        for p_ in fast.model.policy_dicts.parameters():
            p_.zero_()
    acts = []
    for _ in range(2):
        fast.step()
        torch.cuda.synchronize()
        acts.append(_fields(fast)["action"].clone())
    assert (acts[0] - acts[1]).abs().mean().item() > 1e-3 and int(fast.rng_state[1].item()) == 0


def test_safemaddpg_fused_step_applies_the_safety_layer():
    """SAFEMADDPG in the graph rollout: policy kernel -> flexenv_safety_project -> env kernel -> pack, against the
    reference's order of operations (safemaddpg.py:90-111, model.py:211-232) spelled out with the PyTorch module."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
    from train_maddpg import DEFAULT_ALG_ARGS
    from safe_marl_amd.flex_env import VecFlexProvisionEnv
    from safe_marl_amd.learner import SAFEMADDPG, RolloutGraph
    from safe_marl_amd.network import create_network
    from safe_marl_amd.replay_buffer import TransReplayBuffer
    from safe_marl_amd.series import make_synthetic_series
    from safe_marl_amd.util import convert
    net = create_network()
    series = make_synthetic_series(net, n_days=30)
    alg = dict(DEFAULT_ALG_ARGS)
    alg.update(alg="safemaddpg", agent_num=5, obs_size=144, state_size=110, action_dim=4, v_min=0.9, v_max=1.1)
    N = 48
    envs = [VecFlexProvisionEnv({"alg": "safemaddpg"}, N, net=net, series=series, seed=9, warm_start=True) for _ in range(2)]
    # a voltage predictor that makes the layer intervene on part of the batch
    pred = (torch.full((5,), -0.08, dtype=torch.float64), torch.full((5,), -0.05, dtype=torch.float64),
            torch.full((5,), 0.915, dtype=torch.float64))
    models = []
    for env in envs:
        torch.manual_seed(21)
        m = SAFEMADDPG(convert(alg), env, predictor=pred).cuda()
        with torch.no_grad():
            for p in m.policy_dicts.parameters():
                p.mul_(10.0)
        models.append(m)
    rg = RolloutGraph(models[0], envs[0], TransReplayBuffer(N * 8, device="cuda"))
    assert rg.fast and rg.safe
    rg.torch_noise = True
    rg.start_episode(envs[0].reset())
    obs = envs[1].reset().clone()
    hid = torch.zeros(N, 5, 64, device="cuda")
    m = models[1]
    m.fused_inference = False
    hits = 0
    for step in range(3):
        torch.manual_seed(300 + step)
        rg.step()
        torch.manual_seed(300 + step)
        with torch.no_grad():
            noise = torch.randn(N, 5, 4, device="cuda")
            means, _, new_hid = m.policy(obs, last_hid=hid)
            action = torch.tanh(means + noise)
            adjusted = m.safety_layer_optimization(action).to(torch.float32)
            hits += int(m._last_intervened.sum())
            envs[1].step(m.env_action(adjusted), fuse_obs=True, auto_reset=True)
        torch.cuda.synchronize()
        assert (_fields(rg)["action"] - action).abs().max().item() < 2e-4     # the replay keeps the policy's own action
        assert (envs[0].reward - envs[1].reward).abs().max().item() < 1e-4 * max(1.0, envs[1].reward.abs().max().item())
        assert (rg.obs - envs[1].obs).abs().max().item() < 5e-4
        obs = envs[1].obs.clone()
        hid = new_hid * (1.0 - envs[1].done.float()).view(N, 1, 1)
    assert hits > 0                                                           # the layer really acted in this test


@pytest.mark.parametrize("alg,N", [("matd3", 40), ("iddpg", 40), ("matd3", 4096)])
def test_general_graph_body_follows_get_actions(alg, N):
    """MATD3 / IDDPG in the graph rollout: the body calls their own get_actions (agent-summed action selection,
    matd3.py:88-111, iddpg.py:66-71) exactly as the eager vector loop does; record, hand-over and statistics against
    that loop spelled out here, then a captured replay."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
    from train_maddpg import DEFAULT_ALG_ARGS
    from safe_marl_amd import learner
    from safe_marl_amd.flex_env import VecFlexProvisionEnv
    from safe_marl_amd.network import create_network
    from safe_marl_amd.replay_buffer import TransReplayBuffer
    from safe_marl_amd.series import make_synthetic_series
    from safe_marl_amd.util import convert
    net = create_network()
    series = make_synthetic_series(net, n_days=30)
    a = dict(DEFAULT_ALG_ARGS)
    a.update(alg=alg, agent_num=5, obs_size=144, state_size=110, action_dim=4)
    cls = {"matd3": learner.MATD3, "iddpg": learner.IDDPG}[alg]
    envs = [VecFlexProvisionEnv({}, N, net=net, series=series, seed=4, warm_start=True) for _ in range(2)]
    torch.manual_seed(8)
    m = cls(convert(a)).cuda()
    with torch.no_grad():
        for p in m.policy_dicts.parameters():
            p.mul_(10.0)
    rg = learner.RolloutGraph(m, envs[0], TransReplayBuffer(N * 8, device="cuda"))
    assert not rg.fast and not rg.plain
    rg.start_episode(envs[0].reset())
    obs = envs[1].reset().clone()
    hid = torch.zeros(N, 5, 64, device="cuda")
    avail = torch.ones(N, 5, 4, device="cuda")
    for step in range(3):
        torch.manual_seed(500 + step)
        rg.step()
        torch.manual_seed(500 + step)
        with torch.no_grad():
            action, action_pol, _, _, new_hid = m.get_actions(obs, status="train", exploration=True, actions_avail=avail,
                                                              target=False, last_hid=hid)
            envs[1].step(m.env_action(action), fuse_obs=True, auto_reset=True)
        torch.cuda.synchronize()
        f = _fields(rg)
        assert torch.equal(f["state"], obs)
        assert torch.equal(f["action"], action_pol.expand(N, 5, 4))
        assert torch.equal(f["last_hid"], hid)
        assert torch.equal(envs[0].reward, envs[1].reward)
        assert torch.equal(f["next_state"], envs[1].obs)
        obs = envs[1].obs.clone()
        hid = new_hid * (1.0 - envs[1].done.float()).view(N, 1, 1)
        assert torch.equal(f["hid"], hid)                  # model.py:241's hid, zeroed where the episode ended (model.py:255-258)
        assert torch.equal(rg.obs, obs) and torch.equal(rg.hid, hid)
    os.environ["FLEX_GRAPH_AUDIT"] = "1"          # capture() first checks its body for ATen multi-block reductions
    try:
        rg.capture()
    finally:
        del os.environ["FLEX_GRAPH_AUDIT"]
    # round 3: the env step files the transition itself here too (three launches: policy, action selection, environment)
    assert rg.summed_sink and rg.sink_active
    assert not any("rollout_pack_kernel" in k for k in rg.audit) and any("flex_step_kernel" in k for k in rg.audit)
    assert any("agent_sum_explore" in k for k in rg.audit) and any("actor_" in k for k in rg.audit)
    rg.start_episode(envs[0].reset())
    before = rg.obs.clone()
    want_info = torch.zeros_like(rg.info_sum)
    want_rew = torch.zeros_like(rg.rew_sum)
    for _ in range(6):
        rg.step()
        # REPLAYED statistics (the headline batch size included) against eager sums of what the replay left in the env
        want_info += envs[0].info.sum(0)
        want_rew += envs[0].reward.sum()
        torch.cuda.synchronize()
        assert torch.allclose(rg.info_sum, want_info, rtol=1e-12, atol=1e-12), (rg.info_sum, want_info)
        assert torch.allclose(rg.rew_sum, want_rew, rtol=1e-12, atol=1e-12)
        assert rg.fail_sum.item() == 0.0
    assert not torch.equal(before, rg.obs) and all(torch.isfinite(v).all() for v in _fields(rg).values())


def test_graph_replays_equal_eager_steps_bit_for_bit():
    """The captured rollout step against the same body run eagerly, from the same environment state, policy, hidden
    state and noise-stream position: ring contents, hand-over tensors and statistics identical over ten steps (in-launch
    auto-resets excluded by the short horizon; both use the actor kernel's own noise stream)."""
    import numpy as np
    graph, eager = _setup(96)
    graph.start_episode(graph.env.reset())
    graph.capture()                                   # warm-up + capture advance env, hidden state and the noise step
    rng = np.random.default_rng(0)
    n, na = 96, 5
    spec = dict(day=rng.integers(0, 20, n).astype(np.int32), hour=rng.integers(0, 20, n).astype(np.int32),
                interval=rng.integers(0, 4, n).astype(np.int32), e0=0.0125 + 0.001 * rng.random((n, na)),
                a0=0.5 + 0.5 * rng.random((n, 4 * na)))
    for rg in (graph, eager):
        rg.start_episode(rg.env.reset(spec=spec))
        rg.rng_state.copy_(torch.tensor([1234567, 42], dtype=torch.int64))
    assert graph.graph is not None and eager.graph is None
    for step in range(10):
        graph.step()
        eager.step()
        torch.cuda.synchronize()
        for name in ("obs", "hid", "info_sum", "rew_sum", "fail_sum", "rng_state"):
            assert torch.equal(getattr(graph, name), getattr(eager, name)), (step, name)
        fg, fe = _fields(graph), _fields(eager)
        for name in fg:
            assert torch.equal(fg[name], fe[name]), (step, name)
        assert torch.equal(graph.buf.cursor, eager.buf.cursor)
    assert int(graph.rng_state[1]) == 52 and graph.buf.k == 10 and graph.buf.cursor.tolist() == [10 % graph.buf.slabs, 9 % graph.buf.slabs]


def test_ring_wraps_and_windows_stay_consecutive():
    """Slab ring bookkeeping on the device and its host mirror: 73 steps through a ring of 12 + 23 slabs (row mode: the 23
    slabs behind the oldest transition stay, its stacked observation reaches into them), two hard restarts on the way
    (gaps), sampled windows are consecutive transitions of one stream (utils/replay_buffer.py:17-21) whose next_state is
    the following slab's state — every stacked observation formed from the row ring equals what the policy saw — and the
    gathered static batch equals the window read eagerly."""
    import numpy as np
    from safe_marl_amd.replay_buffer import TransReplayBuffer
    from safe_marl_amd.learner import RolloutGraph
    (rg, _), N = _setup(16), 16
    rg = RolloutGraph(rg.model, rg.env, TransReplayBuffer(N * 12, device="cuda"))
    buf = rg.buf
    assert buf.row_mode and buf.keep_back == 23 and buf.slabs == 12 + 23
    S = buf.slabs
    rg.start_episode(rg.env.reset())
    states = {}
    for t in range(73):
        if t in (17, 55):
            rg.start_episode(rg.env.reset())          # hard restart: the slab at the cursor stays half-written -> gap
        k = buf.k
        states[k] = rg.obs.clone()
        rg.step()
    torch.cuda.synchronize()
    K = 75                                            # 73 steps + 2 gap slabs
    assert buf.gaps == [] or all(g >= buf.first for g in buf.gaps)
    assert buf.k == K and buf.cursor.tolist() == [K % S, (K - 1) % S] and buf.first == K + 2 - 12
    assert len(buf.buffer) == N * (K - buf.first - len(buf.gaps))
    np.random.seed(0)
    for _ in range(50):
        bs = 3 * N + 5
        slot = buf.sample_slot(bs)
        j0, j1 = slot // N, (slot + bs - 1) // N
        assert buf.first <= j0 and j1 < buf.k and not any(j0 <= g <= j1 for g in buf.gaps)
        w = buf.slab_window(slot, bs)
        # slot -> (slab counter, env): state is what the policy saw at that step, next_state the next slab's state
        for r in (0, bs // 2, bs - 1):
            j, e = (slot + r) // N, (slot + r) % N
            assert torch.equal(w.state[r], states[j][e])
            if j + 1 in states:
                assert torch.equal(w.next_state[r], states[j + 1][e])
    # one gather launch per kind fills a static batch with exactly that window (seam of the ring included)
    bs = 2 * N
    plan_out = {k: torch.zeros((bs,) + buf.field_shape(k), device="cuda") for k in buf.STORED}
    seam = (K // S) * S                                # a slab counter at physical slab 0
    assert buf.first < seam - 1 and seam + 1 < buf.k
    for slot in (buf.first * N + 3, (buf.k - 3) * N + 1, (seam - 1) * N + 9):    # the third one crosses the seam
        if any(slot // N <= g <= (slot + bs - 1) // N for g in buf.gaps):
            continue
        plan = []
        for k, dst in plan_out.items():
            ring, col0, width, off = buf.field_source(k)
            plan.append((ring, col0, width, off * N, bs, dst.view(bs, -1)))
        buf.gather(plan[:6], slot)
        buf.gather(plan[6:], slot)
        w = buf.slab_window(slot, bs)
        torch.cuda.synchronize()
        for k, dst in plan_out.items():
            assert torch.equal(dst, getattr(w, k).reshape(dst.shape)), (slot, k)


def test_unbounded_actions_take_the_general_body():
    """util.py:66-74: without action_enforcebound the exploration is mean + noise (no tanh); the actor kernel's epilogue is
    tanh(mean + std * noise), so RolloutGraph.fast must be off and the general body (select_action itself) runs."""
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
    alg.update(alg="maddpg", agent_num=5, obs_size=144, state_size=110, action_dim=4, action_enforcebound=False)
    torch.manual_seed(1)
    m = MADDPG(convert(alg)).cuda()
    env = VecFlexProvisionEnv({}, 32, net=net, series=series, seed=5, warm_start=True)
    rg = RolloutGraph(m, env, TransReplayBuffer(32 * 8, device="cuda"))
    assert not rg.fast and not rg.ring_active and not rg.plain
    rg.start_episode(env.reset())
    obs = rg.obs.clone()
    torch.manual_seed(9)
    rg.step()
    torch.manual_seed(9)
    with torch.no_grad():
        means, _, _ = m.policy(obs, last_hid=torch.zeros(32, 5, 64, device="cuda"))
        from torch.distributions.normal import Normal
        want = means + Normal(torch.zeros_like(means), torch.ones_like(means)).rsample()     # util.py:66-74, fixed std of 1
    got = rg.last_transition().action
    assert (got - want).abs().max().item() < 1e-5 and got.abs().max().item() > 1.0          # really unbounded


@pytest.mark.parametrize("fast", [True, False])
def test_bursts_of_steps_equal_single_steps_bit_for_bit(fast):
    """RolloutGraph.run(m): graphs of 16 / 8 / 4 / 2 bodies and single replays against m single-step replays from the
    same state — 31 + 7 + 16 steps through the wrap of an 8-slab ring: every slab of the ring, the hand-over tensors,
    the statistics, the cursors and the noise position identical; recording a burst moves nothing."""
    import numpy as np
    a, b = _setup(64)
    rng = np.random.default_rng(3)
    n, na = 64, 5
    spec = dict(day=rng.integers(0, 20, n).astype(np.int32), hour=rng.integers(0, 20, n).astype(np.int32),
                interval=rng.integers(0, 4, n).astype(np.int32), e0=0.0125 + 0.001 * rng.random((n, na)),
                a0=0.5 + 0.5 * rng.random((n, 4 * na)))
    for rg in (a, b):
        rg.fast = fast
        rg.start_episode(rg.env.reset())
        rg.capture()
        rg.start_episode(rg.env.reset(spec=spec))
        rg.rng_state.copy_(torch.tensor([99, 7], dtype=torch.int64))
        torch.manual_seed(5)                                   # (the general body draws its noise with torch.randn)
    calls = a.env.calls
    for m in (31, 7, 16):
        if not fast:
            torch.manual_seed(m)
        slabs_a = a.run(m)
        if not fast:
            torch.manual_seed(m)
        slabs_b = [b.step() for _ in range(m)]
        torch.cuda.synchronize()
        assert slabs_a == slabs_b and a.buf.k == b.buf.k
        for name in ("obs", "hid", "info_sum", "rew_sum", "fail_sum", "rng_state"):
            assert torch.equal(getattr(a, name), getattr(b, name)), (m, name)
        assert torch.equal(a.buf.cursor, b.buf.cursor)
        assert a.buf.obs_source_ring == b.buf.obs_source_ring
        for ring in (a.buf.obs_source_ring, "hid_ring", "small_ring"):
            assert torch.equal(getattr(a.buf, ring), getattr(b.buf, ring)), (m, ring)
    # b never replayed a burst.  (fast, five agents: run(m) is the fused burst launch of flexenv_rollout_burst, any length
    # below the ring's eight slabs — the comparison above is that launch against single policy + environment launches)
    assert a.fused_burst == fast
    if fast:       # every length is a graph of its own, recorded at first use here (capture() was given no schedule):
        assert a.buf.slabs == 8 + 23                   # row mode: 23 slabs of history behind the 8 of the ring
        assert sorted(k for k, _ in a.bursts) == [7, 16, 30] and not b.bursts        # 31 = 30 + 1 under a 31-slab ring, 7, 16
    else:
        assert sorted(k for k, _ in a.bursts) == [2, 4, 8, 16] == sorted(k for k, _ in b.bursts)
    assert a.env.calls == calls                                # replays never go through env.step; recording is undone


def test_training_loop_schedule_is_unchanged_by_bursts(monkeypatch):
    """train_process with bursts (five agents: the fused burst launch, one launch per run of steps between two update
    events) against the step-by-step loop (BURSTS emptied, FLEX_ROLLOUT_BURST=0): the same update events at the same
    steps — identical parameters, optimiser state and replay ring after two episodes with updates every 7 steps and
    target updates every 11."""
    import numpy as np
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
    from train_maddpg import DEFAULT_ALG_ARGS
    from safe_marl_amd.flex_env import VecFlexProvisionEnv
    from safe_marl_amd.learner import MADDPG, RolloutGraph
    from safe_marl_amd.network import create_network
    from safe_marl_amd.series import make_synthetic_series
    from safe_marl_amd.trainer import PGTrainer
    from safe_marl_amd.util import convert
    net = create_network()
    series = make_synthetic_series(net, n_days=30)
    alg = dict(DEFAULT_ALG_ARGS)
    alg.update(alg="maddpg", agent_num=5, obs_size=144, state_size=110, action_dim=4, behaviour_update_freq=7,
               target_update_freq=11, replay_warmup=0, batch_size=4, value_update_epochs=2, policy_update_epochs=1)
    res = []
    saved = RolloutGraph.BURSTS
    try:
        for bursts in (saved, ()):
            RolloutGraph.BURSTS = bursts
            monkeypatch.setenv("FLEX_ROLLOUT_BURST", "1" if bursts else "0")
            torch.manual_seed(3)
            np.random.seed(3)                                  # replay windows are drawn with numpy (utils/replay_buffer.py:17-21)
            env = VecFlexProvisionEnv({}, 256, net=net, series=series, seed=9, warm_start=True)
            tr = PGTrainer(convert(alg), MADDPG, env, None, replay_capacity=256 * 96 * 2)
            for _ in range(2):
                tr.behaviour_net.train_process({}, tr)
            torch.cuda.synchronize()
            rg = tr.behaviour_net._rollout_graph
            assert bool(rg.bursts) == bool(bursts) == rg.fused_burst
            res.append((tr.steps, [p.detach().clone() for p in tr.behaviour_net.parameters()],
                        [p.detach().clone() for p in tr.behaviour_net.target_net.parameters()],
                        tr.replay_buffer.small_ring.clone(), getattr(tr.replay_buffer, tr.replay_buffer.obs_source_ring).clone()))
    finally:
        RolloutGraph.BURSTS = saved
    assert res[0][0] == res[1][0] == 190
    for x, y in zip(res[0][1] + res[0][2] + list(res[0][3:]), res[1][1] + res[1][2] + list(res[1][3:])):
        assert torch.equal(x, y)


@pytest.mark.parametrize("alg", ["matd3", "iddpg"])
def test_one_launch_action_selection_equals_get_actions(alg):
    """MATD3 / IDDPG.get_actions(need_log_prob=False) on the constant availability mask: exploration (tanh of the
    agent-summed mean plus noise) and the no-exploration summed mean, each from one launch of flexnet_agent_sum_explore,
    against the reference-shaped tensor composition — actions bit for bit (same draws), and the gradient w.r.t. the
    policy's parameters through the summed mean."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
    from train_maddpg import DEFAULT_ALG_ARGS
    from safe_marl_amd import learner
    from safe_marl_amd.util import convert
    a = dict(DEFAULT_ALG_ARGS)
    a.update(alg=alg, agent_num=5, obs_size=144, state_size=110, action_dim=4)
    cls = {"matd3": learner.MATD3, "iddpg": learner.IDDPG}[alg]
    torch.manual_seed(3)
    m = cls(convert(a)).cuda()
    with torch.no_grad():
        for p in m.policy_dicts.parameters():
            p.mul_(10.0)
    b = 4096
    g = torch.Generator(device="cuda").manual_seed(1)
    obs = torch.randn(b, 5, 144, device="cuda", generator=g)
    hid = torch.randn(b, 5, 64, device="cuda", generator=g)
    const = torch.ones(1, 1, 1, device="cuda").expand(b, 5, 4)
    const._flex_const = 1.0
    real = torch.ones(b, 5, 4, device="cuda")                     # a materialised mask takes the tensor composition
    with torch.no_grad():
        torch.manual_seed(77)
        fa, fr, flp, _, fh = m.get_actions(obs, "train", True, const, last_hid=hid, need_log_prob=False)
        torch.manual_seed(77)
        sa, sr, slp, _, sh = m.get_actions(obs, "train", True, real, last_hid=hid)
    assert flp is None and slp is not None
    assert torch.equal(fa, sa) and torch.equal(fr, sr.expand(b, 5, 4)) and torch.equal(fh, sh)
    # no exploration, with gradients (the policy loss's evaluation)
    outs = []
    for mask in (const, real):
        act, rest, _, _, _ = m.get_actions(obs, "train", False, mask, last_hid=hid, need_log_prob=False)
        w = torch.linspace(-1, 1, b * 20, device="cuda").view(b, 5, 4)
        grads = torch.autograd.grad((rest * w).sum(), list(m.policy_dicts.parameters()))
        outs.append((act.detach(), rest.detach().expand(b, 5, 4).clone(), grads))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    for x, y in zip(outs[0][2], outs[1][2]):
        assert (x - y).abs().max().item() <= 2e-4 * max(1e-6, y.abs().max().item())


def _trainer(n_envs, buildings=None, **over):
    import numpy as np
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
    from train_maddpg import DEFAULT_ALG_ARGS
    from safe_marl_amd.flex_env import VecFlexProvisionEnv
    from safe_marl_amd.learner import MADDPG
    from safe_marl_amd.network import create_network
    from safe_marl_amd.series import make_synthetic_series
    from safe_marl_amd.trainer import PGTrainer
    from safe_marl_amd.util import convert
    env_args = {} if buildings is None else {"buildings": buildings, "pv_nodes": buildings, "ess_nodes": buildings}
    net = create_network(env_args)
    series = make_synthetic_series(net, n_days=30)
    env = VecFlexProvisionEnv(env_args, n_envs, net=net, series=series, seed=9, warm_start=True)
    alg = dict(DEFAULT_ALG_ARGS)
    alg.update(alg="maddpg", agent_num=env.n_agents, obs_size=env.obs_size, state_size=env.state_size, action_dim=4,
               behaviour_update_freq=30, batch_size=4)
    alg.update(over)
    torch.manual_seed(3)
    np.random.seed(3)
    return PGTrainer(convert(alg), MADDPG, env, None, replay_capacity=n_envs * 96 * 2)


def test_capture_failure_falls_back_to_the_eager_rollout(monkeypatch):
    """a failed rollout-graph capture used to leave the replay buffer in slab mode and the env's device-side
    hooks pointed at the ring, so the promised eager rollout raised in add_batch.  Now the ring set-up is undone."""
    import warnings
    from safe_marl_amd.learner import RolloutGraph
    tr = _trainer(256)

    def boom(self):
        raise RuntimeError("forced capture failure")

    monkeypatch.setattr(RolloutGraph, "capture", boom)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        for _ in range(2):
            stat = {}
            tr.behaviour_net.train_process(stat, tr)
    assert any("rollout graph capture failed" in str(x.message) for x in w)
    assert tr.graph_rollout is False and not tr.replay_buffer.slab_mode
    assert tr.steps == 190 and len(tr.replay_buffer.buffer) == 190 * 256
    assert all(torch.isfinite(torch.as_tensor(float(v))) for v in stat.values())
    assert "mean_train_value_loss" in stat                       # update events ran on the field-by-field buffer


def test_seven_agents_take_the_pack_path(monkeypatch):
    """7 agents x 64 hidden units = 448 floats per env exceed the env-filed transition's 384-float recurrent
    row (flexenv_set_replay_sink); the rollout must pick the three-launch body (pack kernel) instead of raising."""
    blds = [5, 8, 10, 15, 20, 25, 30]
    tr = _trainer(128, buildings=blds)
    for _ in range(2):
        stat = {}
        tr.behaviour_net.train_process(stat, tr)
    rg = tr.behaviour_net._rollout_graph
    assert rg.graph is not None and rg.fast and rg.ring_active and not rg.sink and not rg.sink_active
    assert tr.env.n_agents == 7 and tr.replay_buffer.slab_mode and tr.steps == 190
    assert sorted(tr._update_graphs) == ["policy", "value"]
    assert all(torch.isfinite(torch.as_tensor(float(v))) for v in stat.values())
    # the ring holds what the policy did: actions of the last slab are tanh-bounded, rewards finite, 7 agents wide
    t = rg.last_transition()
    assert t.action.shape == (128, 7, 4) and t.action.abs().max().item() <= 1.0 and torch.isfinite(t.reward).all()


@pytest.mark.parametrize("n_envs,buildings", [(4100, None), (531, [3, 17, 28]), (77, [5]), (203, [3, 9, 17, 28])])
def test_burst_launch_equals_two_launches_per_step_bit_for_bit(n_envs, buildings, monkeypatch):
    """flexenv_rollout_burst (policy + environment for m steps in one persistent launch, weights staged once per CU)
    against the same trainer with FLEX_ROLLOUT_BURST=0 (policy launch + environment launch per step): a batch that is not a
    multiple of a block's sixteen environments (tail block, empty second group), more blocks than CUs (4100 -> 257), one /
    three / four agents (8 / 24 / 32 policy rows per group: one or two tiles, the last one partly filled); 40 + 23 + 60 steps with the episodes' restart inside the last burst.  Every
    ring, cursor, statistic, the noise position and the environments' own state afterwards: identical bits."""
    import numpy as np
    from safe_marl_amd import nets
    from safe_marl_amd.learner import RolloutGraph
    runs = []
    # the two-launch side on the 16-row-tile policy kernel whatever the batch (above 80 rows per CU the library would pick
    # its 32-row kernel, whose LayerNorm sums are grouped differently: equal to 2e-6, not to the bit)
    monkeypatch.setattr(nets, "ACTOR_VARIANT", 2)
    for flag in ("1", "0"):
        monkeypatch.setenv("FLEX_ROLLOUT_BURST", flag)
        tr = _trainer(n_envs, buildings)
        m = tr.behaviour_net
        with torch.no_grad():
            for p in m.policy_dicts.parameters():
                p.mul_(10.0)
        rg = RolloutGraph(m, tr.env, tr.replay_buffer)
        assert rg.fused_burst == (flag == "1") and rg.sink_active
        rng = np.random.default_rng(4)
        na = tr.env.n_agents
        spec = dict(day=rng.integers(0, 20, n_envs).astype(np.int32), hour=rng.integers(0, 24, n_envs).astype(np.int32),
                    interval=rng.integers(0, 4, n_envs).astype(np.int32), e0=0.0125 + 0.001 * rng.random((n_envs, na)),
                    a0=0.5 + 0.5 * rng.random((n_envs, 4 * na)))
        rg.start_episode(tr.env.reset())
        rg.capture()
        rg.start_episode(tr.env.reset(spec=spec))
        rg.rng_state.copy_(torch.tensor([1234, 5], dtype=torch.int64))
        slabs = rg.run(40) + rg.run(23) + rg.run(60)         # (every episode ends at step 96: restarts inside the third burst)
        torch.cuda.synchronize()
        runs.append((rg, slabs, tr))
    (a, sa, ta), (b, sb, tb) = runs
    assert sa == sb and a.buf.k == b.buf.k
    assert torch.equal(a.buf.cursor, b.buf.cursor) and torch.equal(a.rng_state, b.rng_state)
    assert int(a.rng_state[1].item()) == 5 + 123
    assert a.buf.row_mode and b.buf.row_mode             # the env's step files feature rows; nothing copies an observation
    for ring in ("row_ring", "hid_ring", "small_ring"):
        assert torch.equal(getattr(a.buf, ring), getattr(b.buf, ring)), ring
    for name in ("acc", "act_buf", "hid_buf"):
        assert torch.equal(getattr(a, name), getattr(b, name)), name
    for name in ("reward", "done", "info", "failed"):
        assert torch.equal(getattr(ta.env, name), getattr(tb.env, name)), name
    assert torch.equal(ta.env.obs_view().clone(), tb.env.obs_view().clone())
    assert float(a.acc[:, 7].abs().sum().item()) > 0


@pytest.mark.parametrize("N", [8200, 75])
def test_safemaddpg_burst_equals_three_launches_per_step_bit_for_bit(N, monkeypatch):
    """SAFEMADDPG through flexenv_rollout_burst (the safety projection as a phase of the persistent launch: each wavefront
    projects its own two environments between the policy and the step) against policy launch + flexenv_safety_project_env +
    environment launch per step, with a voltage predictor that makes the layer intervene: 8200 environments (513 blocks: two
    rounds of persistent blocks and a tail block) and 75; 40 + 70 steps through the episodes' restart.  Identical bits in every
    ring, cursor, statistic and in the environments' state."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
    from train_maddpg import DEFAULT_ALG_ARGS
    from safe_marl_amd import nets
    from safe_marl_amd.flex_env import VecFlexProvisionEnv
    from safe_marl_amd.learner import SAFEMADDPG, RolloutGraph
    from safe_marl_amd.network import create_network
    from safe_marl_amd.replay_buffer import TransReplayBuffer
    from safe_marl_amd.series import make_synthetic_series
    from safe_marl_amd.util import convert
    net = create_network()
    series = make_synthetic_series(net, n_days=30)
    alg = dict(DEFAULT_ALG_ARGS)
    alg.update(alg="safemaddpg", agent_num=5, obs_size=144, state_size=110, action_dim=4, v_min=0.9, v_max=1.1)
    pred = (torch.full((5,), -0.08, dtype=torch.float64), torch.full((5,), -0.05, dtype=torch.float64),
            torch.full((5,), 0.915, dtype=torch.float64))
    monkeypatch.setattr(nets, "ACTOR_VARIANT", 2)            # (the two-launch side on the 16-row policy kernel at every size)
    runs = []
    for flag in ("1", "0"):
        monkeypatch.setenv("FLEX_ROLLOUT_BURST", flag)
        env = VecFlexProvisionEnv({"alg": "safemaddpg"}, N, net=net, series=series, seed=9, warm_start=True)
        torch.manual_seed(21)
        m = SAFEMADDPG(convert(alg), env, predictor=pred).cuda()
        with torch.no_grad():
            for p in m.policy_dicts.parameters():
                p.mul_(10.0)
        rg = RolloutGraph(m, env, TransReplayBuffer(N * 128, device="cuda"))
        assert rg.safe and rg.sink_active and rg.fused_burst == (flag == "1")
        rg.start_episode(env.reset())
        rg.capture()
        rg.start_episode(env.reset())
        rg.rng_state.copy_(torch.tensor([77, 3], dtype=torch.int64))
        slabs = rg.run(40) + rg.run(70)
        torch.cuda.synchronize()
        runs.append((rg, slabs, env))
    (a, sa, ea), (b, sb, eb) = runs
    assert sa == sb and torch.equal(a.buf.cursor, b.buf.cursor) and torch.equal(a.rng_state, b.rng_state)
    assert a.buf.row_mode and b.buf.row_mode             # the env's step files feature rows; nothing copies an observation
    for ring in ("row_ring", "hid_ring", "small_ring"):
        assert torch.equal(getattr(a.buf, ring), getattr(b.buf, ring)), ring
    for name in ("acc", "act_buf", "hid_buf"):
        assert torch.equal(getattr(a, name), getattr(b, name)), name
    for name in ("reward", "done", "info", "failed"):
        assert torch.equal(getattr(ea, name), getattr(eb, name)), name
    assert torch.equal(ea.obs_view().clone(), eb.obs_view().clone())
    # the layer did act: the step's action differs from translate_action of the policy's own action somewhere
    own = 0.5 * (a.act_buf.clamp(0.0, 1.0) + 1.0)
    assert (a.burst_safe_env_act.view(N, 4, 5).transpose(1, 2).reshape(N * 5, 4) - own).abs().max().item() > 1e-3


def test_burst_entry_refuses_what_it_cannot_run(monkeypatch):
    """flexenv_rollout_burst's argument checks (include/flexenv.h): a burst as long as the ring, an env without the replay
    sink, a safety block with another action width — FLEX_EINVAL, nothing launched, the state untouched."""
    from safe_marl_amd import _lib
    from safe_marl_amd.learner import RolloutGraph
    tr = _trainer(64)
    rg = RolloutGraph(tr.behaviour_net, tr.env, tr.replay_buffer)
    assert rg.fused_burst
    rg.start_episode(tr.env.reset())
    rg.capture()
    rg.run(5)
    torch.cuda.synchronize()
    cursor, calls = rg.buf.cursor.clone(), tr.env.calls
    with pytest.raises(_lib.FlexLibraryError, match="flexenv_rollout_burst failed with code -22"):      # FLEX_EINVAL
        rg.body(burst=rg.buf.slabs)                           # steps >= slabs
    tr.env.calls = calls
    tr.env.set_replay_sink(None, None, None, None, None)      # sink off
    with pytest.raises(_lib.FlexLibraryError, match="code -22"):
        rg.body(burst=4)
    tr.env.calls = calls
    rg._configure_env()                                       # sink back on: the next burst runs
    torch.cuda.synchronize()
    assert torch.equal(rg.buf.cursor, cursor)
    before = rg.buf.k
    rg.run(6)
    torch.cuda.synchronize()
    assert rg.buf.k == before + 6
