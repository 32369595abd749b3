"""GPU: the PRODUCT update path (fused HIP kernels on) against the golden vectors captured by importing the
reference's own modules (tests/golden/make_learner_golden.py) — the one-hop check reference -> HIP path.

Reference symbols pinned here: madrl/models/maddpg.py:100-123 (get_loss), madrl/models/model.py:102-140,308-323 (policy,
unpack_data / reward BatchNorm), utils/trainer.py:81-108 (zero_grad -> backward -> clip_grad_norm_ -> RMSprop),
madrl/models/model.py:28-38 (update_target), madrl/models/matd3.py:111-149, madrl/models/iddpg.py + learning_algorithms/ddpg.py.

The golden batch has 32 samples.  The kernels switch implementation with the row count (VALU kernels below 65 536
critic rows, matrix-core kernels above; csrc/wgrad.hip / csrc/lnrelu.hip from 2 048 rows), so the batch is also TILED
k times: every loss of the path is a mean over samples and the reward BatchNorm uses biased batch statistics, so
losses, gradients, the RMSprop step and the target update are invariant under tiling (only the BatchNorm's
running_var sees the unbiased n/(n-1) factor, which the test accounts for).  Tolerances are those of the CPU test
(tests/test_learner_cpu.py): losses 1e-5, gradients 2e-6 + 1e-4 max|g|, post-step weights 3e-6 (+ the RMSprop
sensitivity of near-zero gradients, stated below)."""
import json
import os

import numpy as np
import pytest
import torch as th

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")

# 1: 32 samples (small-batch kernels); 64: 2 048 samples = 10 240 actor rows (wgrad + lnrelu + VALU critic tail);
# 2048: 65 536 samples = 327 680 rows (matrix-core critic tail / pgrad kernels, MFMA actor at full width)
TILES = [1, 64, 2048]


@pytest.fixture(scope="module")
def gold():
    return dict(np.load(os.path.join(G, "learner_golden.npz")))


def _args(prefix="learner", **over):
    from safe_marl_amd.util import convert
    d = json.load(open(os.path.join(G, prefix + "_args.json")))
    d.update(cuda=True)
    d.update(over)
    return convert(d)


def _load_sd(name):
    z = np.load(os.path.join(G, name))
    return {k: th.from_numpy(z[k]) for k in z.files}


def _batch(tile=1, prefix="learner"):
    from safe_marl_amd.replay_buffer import Transition
    z = np.load(os.path.join(G, prefix + "_batch.npz"))
    out = {}
    for k in Transition._fields:
        t = th.from_numpy(z[k]).float().cuda()
        out[k] = t.repeat((tile,) + (1,) * (t.dim() - 1)).contiguous()
    return Transition(**out)


class StubEnv:
    n_envs = 1

    def __init__(self, n=5):
        self.n = n

    def get_num_of_agents(self):
        return self.n


def _model(cls, sd_name, prefix="learner"):
    args = _args(prefix)
    model = cls(args, cls(args).cuda()).cuda()
    res = model.load_state_dict(_load_sd(sd_name), strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    return model


def _np(t):
    return t.detach().float().cpu().numpy()


def _grads(loss, params):
    """what trainer._sub_update hands the optimiser: d loss / d (this optimiser's parameters)"""
    return [_np(g) for g in th.autograd.grad(loss, list(params), allow_unused=False)]


def _loaded_lib():
    from safe_marl_amd import _lib
    return _lib.load()


@pytest.mark.parametrize("tile", TILES)
def test_maddpg_forward_losses_and_grads_match_the_reference(gold, tile):
    """maddpg.py:33-123 on the GPU: policy(), value(), both losses and every parameter gradient."""
    from safe_marl_amd.learner import MADDPG
    _loaded_lib()
    b = _batch(tile)
    model = _model(MADDPG, "learner_state_dict.npz")
    # reward BatchNorm of unpack_data (model.py:321-322): biased batch statistics -> tiling-invariant
    assert np.allclose(_np(model.unpack_data(b)[5])[:32], gold["unpack_reward_bn"], atol=2e-5)
    model = _model(MADDPG, "learner_state_dict.npz")
    with th.no_grad():                                   # fused inference kernel (csrc/actor.hip)
        means, _, hid = model.policy(b.state, last_hid=b.last_hid)
    assert np.allclose(_np(means)[:32], gold["policy_means"], atol=5e-6)
    assert np.allclose(_np(hid)[:32], gold["policy_hiddens"], atol=5e-6)
    means_g, _, hid_g = model.policy(b.state, last_hid=b.last_hid)          # update pass (autograd graph recorded)
    assert np.allclose(_np(means_g)[:32], gold["policy_means"], atol=5e-6)
    assert np.allclose(_np(hid_g)[:32], gold["policy_hiddens"], atol=5e-6)
    with th.no_grad():
        v = model.value(b.state, b.action)
    assert np.allclose(_np(v)[:32], gold["value_sa"], atol=2e-5)
    v_g = model.value(b.state, b.action)                                   # with graph: _CriticReplayedFn at >= 2048 rows
    assert np.allclose(_np(v_g)[:32], gold["value_sa"], atol=2e-5)

    # the reference's call (both losses), then the single-loss evaluations the trainer uses
    model = _model(MADDPG, "learner_state_dict.npz")
    pl, vl, _ = model.get_loss(b)
    assert abs(pl.item() - gold["policy_loss"]) < 1e-5
    assert abs(vl.item() - gold["value_loss"]) < 1e-5 * max(1.0, abs(gold["value_loss"]))
    model = _model(MADDPG, "learner_state_dict.npz")
    _, vl, _ = model.get_loss(b, need="value")           # csrc/tdloss.hip, critic kernels, fused bootstrap actor
    assert abs(vl.item() - gold["value_loss"]) < 1e-5 * max(1.0, abs(gold["value_loss"]))
    names = [k for k, _ in model.value_dicts.named_parameters()]
    for k, g in zip(names, _grads(vl, model.value_dicts.parameters())):
        ref = gold["vgrad." + k]
        assert np.allclose(g, ref, atol=2e-6 + 1e-4 * np.abs(ref).max()), (tile, k, np.abs(g - ref).max())
    pl, _, _ = model.get_loss(b, need="policy")          # lnrelu / wgrad / dz1-only critic backward
    assert abs(pl.item() - gold["policy_loss"]) < 1e-5
    names = [k for k, _ in model.policy_dicts.named_parameters()]
    for k, g in zip(names, _grads(pl, model.policy_dicts.parameters())):
        ref = gold["pgrad." + k]
        assert np.allclose(g, ref, atol=2e-7 + 1e-4 * np.abs(ref).max()), (tile, k, np.abs(g - ref).max())


def _expected_running_var(rv_gold, n_rows):
    """Two training-mode BatchNorm updates from running_var = 1 with momentum 0.1 (one per get_loss call, model.py:308-323):
    0.81 + 0.19 u with u = unbiased batch variance.  The golden value has n = 32; a tiled batch has the same BIASED
    variance, i.e. u_n = u_32 * (31/32) * n/(n-1)."""
    u32 = (rv_gold - 0.81) / 0.19
    return 0.81 + 0.19 * u32 * (31.0 / 32.0) * n_rows / (n_rows - 1.0)


def _check_after_step(gold, trainer, stat, n_rows, label, fixtures="learner"):
    args = trainer.args
    for k in ("mean_train_value_grad_norm", "mean_train_value_loss", "mean_train_policy_grad_norm",
              "mean_train_policy_loss", "mean_train_entropy"):
        assert abs(float(stat[k]) - gold["stat." + k]) < 1e-4 * max(1.0, abs(gold["stat." + k])), (label, k)
    after = _load_sd(fixtures + "_state_dict_after_step.npz")
    before = _load_sd(fixtures + "_state_dict.npz")
    mine = {k: v.detach().cpu() for k, v in trainer.behaviour_net.state_dict().items()}
    # RMSprop's first step is lr * g / (0.1 |g| + eps): for |g| >> 10 eps it is +-10 lr whatever g is, for a near-zero
    # gradient it moves by (lr eps / (0.1 |g| + eps)^2) per unit of gradient error.  Tolerance = 3e-6 (the CPU test's) +
    # that sensitivity times the gradient tolerance of the test above.
    worst = 0.0
    for which, prefix, lr in (("value", "value_dicts.", args.value_lrate), ("policy", "policy_dicts.", args.policy_lrate)):
        norm = float(gold[f"stat.mean_train_{which}_grad_norm"])
        clip = min(1.0, args.grad_clip_eps / (norm + 1e-6))
        gkey = "vgrad." if which == "value" else "pgrad."
        for k, ref in after.items():
            if not k.startswith(prefix):
                continue
            g = th.from_numpy(gold[gkey + k[len(prefix):]]) * clip
            dg = (2e-6 + 1e-4 * g.abs().max()) if g.numel() else 0.0
            sens = lr * 1e-5 / (0.1 * g.abs() + 1e-5) ** 2
            tol = 3e-6 + 1e-5 * ref.abs() + sens * dg
            err = (mine[k] - ref).abs()
            assert bool((err <= tol).all()), (label, k, float((err - tol).max()))
            worst = max(worst, float(err.max()))
            assert not th.equal(mine[k], before[k]), (label, k)          # the step really moved this tensor
    assert mine["batchnorm.num_batches_tracked"].item() == after["batchnorm.num_batches_tracked"].item() == 2
    assert th.allclose(mine["batchnorm.running_mean"], after["batchnorm.running_mean"], atol=1e-6)
    rv = _expected_running_var(after["batchnorm.running_var"].double(), n_rows).float()
    assert th.allclose(mine["batchnorm.running_var"], rv, atol=1e-6, rtol=1e-5), (label, mine["batchnorm.running_var"], rv)
    for k, ref in after.items():                                           # the target replica is untouched by a step
        if k.startswith("target_net.") and ref.is_floating_point() and "batchnorm" not in k:
            assert th.equal(mine[k], before[k]), (label, k)
    return worst


@pytest.mark.parametrize("tile", TILES)
def test_maddpg_one_optimizer_step_and_target_update_match_the_reference(gold, tile):
    """utils/trainer.py:81-108 (value then policy sub-update through PGTrainer: fused losses, backward kernels,
    csrc/optim.hip clip + RMSprop) and model.py:28-38 (soft target update) on the GPU."""
    from safe_marl_amd.learner import MADDPG
    from safe_marl_amd.trainer import PGTrainer
    _loaded_lib()
    trainer = PGTrainer(_args(), MADDPG, StubEnv(), None)
    assert trainer.device.type == "cuda"
    trainer.behaviour_net.load_state_dict(_load_sd("learner_state_dict.npz"))
    b = _batch(tile)
    stat = {}
    trainer.value_transition_process(stat, b)
    trainer.policy_transition_process(stat, b)
    _check_after_step(gold, trainer, stat, 32 * tile, f"eager x{tile}")
    trainer.behaviour_net.update_target()
    tgt = _load_sd("learner_target_after_update.npz")
    mine_t = trainer.behaviour_net.target_net.state_dict()
    for k, ref in tgt.items():
        if ref.is_floating_point() and "batchnorm" not in k:
            # theta' <- 0.9 theta' + 0.1 theta: a tenth of the step tolerance on top of fp32 rounding
            assert th.allclose(mine_t[k].cpu(), ref, atol=3e-6, rtol=1e-5), k


def test_maddpg_graphed_sub_updates_match_the_reference(gold):
    """The same two sub-updates as HIP-graph replays out of the slab replay ring (trainer._graphed_sub_update): the ring
    holds exactly one complete vector step of bs "environments" — slab 0 = (state, last_hid, action, reward, done) of the
    golden batch tiled 64 times, slab 1 = (next_state, hid) — so the only window there is to sample is that batch."""
    from safe_marl_amd.learner import MADDPG
    from safe_marl_amd.trainer import PGTrainer
    _loaded_lib()
    tile = 64
    bs = 32 * tile
    args = _args()
    trainer = PGTrainer(args, MADDPG, StubEnv(), None, batch_scale=tile, replay_capacity=4 * bs, graph_updates=True)
    trainer.behaviour_net.load_state_dict(_load_sd("learner_state_dict.npz"))
    buf = trainer.replay_buffer
    n, o, a, h = 5, 144, 4, 64
    b = _batch(tile)
    buf.alloc_slabs(bs, n, o, a, h)
    buf.begin_stream(b.state)
    buf.hid_ring[0].copy_(b.last_hid.reshape(bs, -1))
    small = buf.small_ring[0]
    small[:, :n * a].copy_(b.action.reshape(bs, -1))
    small[:, n * a:n * a + n].copy_(b.reward)
    small[:, n * a + n].copy_(b.done)
    small[:, n * a + n + 1].copy_(b.last_step)
    buf.obs_ring[1].copy_(b.next_state.reshape(bs, -1))
    buf.hid_ring[1].copy_(b.hid.reshape(bs, -1))
    buf.cursor[0] = 1
    assert buf.stepped() == 0
    assert len(buf.buffer) == bs == trainer.effective_batch_size()
    w = buf.get_batch_tensors(bs)
    for k in ("state", "action", "reward", "next_state", "done", "last_hid", "hid"):
        assert th.equal(getattr(w, k).reshape(getattr(b, k).shape), getattr(b, k)), k
    stat = {}
    trainer.value_replay_process(stat)
    trainer.policy_replay_process(stat)
    th.cuda.synchronize()
    assert trainer.graph_updates and set(trainer._update_graphs) == {"value", "policy"}      # the graph path really ran
    _check_after_step(gold, trainer, stat, bs, "graphed x64")


def _cpu_noise_for(tile):
    """matd3.py:136-138 draws the target-policy smoothing noise from the default CPU generator in the golden run
    (th.manual_seed(99), one draw of the [32, 1, 4] agent-summed shape).  On the GPU the same numbers are handed to
    Normal.rsample, tiled like the batch."""
    import torch.distributions.normal as tdn
    real = tdn._standard_normal

    def fake(shape, dtype, device):
        shape = tuple(shape)
        cpu = real((shape[0] // tile,) + shape[1:], dtype=dtype, device=th.device("cpu"))
        return cpu.repeat((tile,) + (1,) * (len(shape) - 1)).to(device)

    return tdn, real, fake


@pytest.mark.parametrize("tile", [1, 64])
def test_matd3_matches_the_reference_on_the_gpu(gold, tile, monkeypatch):
    """matd3.py:33-149: twin-flag critic values, clipped-double-Q target (min at matd3.py:140), both losses and all
    gradients, with the GPU branches of nets.py (wide_batch_linear, forward_update, fused critic tail) active."""
    from safe_marl_amd.learner import MATD3
    _loaded_lib()
    tdn, real, fake = _cpu_noise_for(tile)
    monkeypatch.setattr(tdn, "_standard_normal", fake)
    model = _model(MATD3, "matd3_state_dict.npz")
    b = _batch(tile)
    v = model.value(b.state, b.action)
    ref = gold["matd3_value"]                               # cat([Q1, Q2]) over 32 samples
    got = _np(v)
    assert got.shape == (64 * tile, 5, 1)
    assert np.allclose(got[:32], ref[:32], atol=2e-5) and np.allclose(got[32 * tile:32 * tile + 32], ref[32:], atol=2e-5)
    th.manual_seed(99)
    pl, vl, _ = model.get_loss(b)
    assert abs(pl.item() - gold["matd3_policy_loss"]) < 1e-5
    assert abs(vl.item() - gold["matd3_value_loss"]) < 1e-5 * max(1.0, abs(gold["matd3_value_loss"]))
    # the trainer's single-loss evaluations
    th.manual_seed(99)
    _, vl, _ = model.get_loss(b, need="value")
    assert abs(vl.item() - gold["matd3_value_loss"]) < 1e-5 * max(1.0, abs(gold["matd3_value_loss"]))
    names = [k for k, _ in model.value_dicts.named_parameters()]
    for k, g in zip(names, _grads(vl, model.value_dicts.parameters())):
        r = gold["matd3_vgrad." + k]
        assert np.allclose(g, r, atol=2e-6 + 1e-4 * np.abs(r).max()), (tile, k, np.abs(g - r).max())
    pl, _, _ = model.get_loss(b, need="policy")
    assert abs(pl.item() - gold["matd3_policy_loss"]) < 1e-5
    names = [k for k, _ in model.policy_dicts.named_parameters()]
    for k, g in zip(names, _grads(pl, model.policy_dicts.parameters())):
        r = gold["matd3_pgrad." + k]
        assert np.allclose(g, r, atol=2e-7 + 1e-4 * np.abs(r).max()), (tile, k, np.abs(g - r).max())


@pytest.mark.parametrize("tile", [1, 64])
def test_iddpg_matches_the_reference_on_the_gpu(gold, tile):
    """iddpg.py:32-83 + learning_algorithms/ddpg.py:14-37 on the GPU."""
    from safe_marl_amd.learner import IDDPG
    _loaded_lib()
    model = _model(IDDPG, "iddpg_state_dict.npz")
    b = _batch(tile)
    v = model.value(b.state, b.action)
    assert np.allclose(_np(v)[:32], gold["iddpg_value"], atol=2e-5)
    pl, vl, _ = model.get_loss(b)
    assert abs(pl.item() - gold["iddpg_policy_loss"]) < 1e-5
    assert abs(vl.item() - gold["iddpg_value_loss"]) < 1e-5 * max(1.0, abs(gold["iddpg_value_loss"]))
    _, vl, _ = model.get_loss(b, need="value")
    names = [k for k, _ in model.value_dicts.named_parameters()]
    for k, g in zip(names, _grads(vl, model.value_dicts.parameters())):
        r = gold["iddpg_vgrad." + k]
        assert np.allclose(g, r, atol=2e-6 + 1e-4 * np.abs(r).max()), (tile, k, np.abs(g - r).max())
    pl, _, _ = model.get_loss(b, need="policy")
    names = [k for k, _ in model.policy_dicts.named_parameters()]
    for k, g in zip(names, _grads(pl, model.policy_dicts.parameters())):
        r = gold["iddpg_pgrad." + k]
        assert np.allclose(g, r, atol=2e-7 + 1e-4 * np.abs(r).max()), (tile, k, np.abs(g - r).max())


# ---- BASELINE.json config 3: MADDPG with THREE agents (critic input (obs + act) * 3 + 3, maddpg.py:18-27) --------------
# fixtures: tests/golden/learner3_* = make_learner_golden.py --agents 3 (the reference's own modules, imported)
@pytest.fixture(scope="module")
def gold3():
    return dict(np.load(os.path.join(G, "learner3_golden.npz")))


@pytest.mark.parametrize("tile", TILES)
def test_maddpg_3_agents_losses_grads_step_and_target_match_the_reference(gold3, tile):
    """maddpg.py:18-27,33-123 + utils/trainer.py:81-108 + model.py:28-38 with agent_num = 3 on the HIP path: values, both
    losses, every parameter gradient, one value + one policy sub-update, the soft target update."""
    from safe_marl_amd.learner import MADDPG
    from safe_marl_amd.trainer import PGTrainer
    from safe_marl_amd import util
    _loaded_lib()
    util.FALLBACKS.clear()                                  # (other tests of the session decline shapes on purpose)
    gold, P = gold3, "learner3"
    b = _batch(tile, P)
    assert b.state.shape[1:] == (3, 144) and b.action.shape[1:] == (3, 4)
    model = _model(MADDPG, P + "_state_dict.npz", P)
    assert model.value_dicts[0].fc1.weight.shape == (64, (144 + 4) * 3 + 3)
    assert np.allclose(_np(model.unpack_data(b)[5])[:32], gold["unpack_reward_bn"], atol=2e-5)
    model = _model(MADDPG, P + "_state_dict.npz", P)
    with th.no_grad():
        means, _, hid = model.policy(b.state, last_hid=b.last_hid)
        v = model.value(b.state, b.action)
    assert np.allclose(_np(means)[:32], gold["policy_means"], atol=5e-6)
    assert np.allclose(_np(hid)[:32], gold["policy_hiddens"], atol=5e-6)
    assert np.allclose(_np(v)[:32], gold["value_sa"], atol=2e-5)
    _, vl, _ = model.get_loss(b, need="value")
    assert abs(vl.item() - gold["value_loss"]) < 1e-5 * max(1.0, abs(gold["value_loss"]))
    names = [k for k, _ in model.value_dicts.named_parameters()]
    for k, g in zip(names, _grads(vl, model.value_dicts.parameters())):
        ref = gold["vgrad." + k]
        assert np.allclose(g, ref, atol=2e-6 + 1e-4 * np.abs(ref).max()), (tile, k, np.abs(g - ref).max())
    pl, _, _ = model.get_loss(b, need="policy")
    assert abs(pl.item() - gold["policy_loss"]) < 1e-5
    names = [k for k, _ in model.policy_dicts.named_parameters()]
    for k, g in zip(names, _grads(pl, model.policy_dicts.parameters())):
        ref = gold["pgrad." + k]
        assert np.allclose(g, ref, atol=2e-7 + 1e-4 * np.abs(ref).max()), (tile, k, np.abs(g - ref).max())
    # one optimiser step of each kind through the trainer, then the target update
    trainer = PGTrainer(_args(P), MADDPG, StubEnv(3), None)
    trainer.behaviour_net.load_state_dict(_load_sd(P + "_state_dict.npz"))
    stat = {}
    trainer.value_transition_process(stat, b)
    trainer.policy_transition_process(stat, b)
    _check_after_step(gold, trainer, stat, 32 * tile, f"3 agents x{tile}", P)
    trainer.behaviour_net.update_target()
    tgt = _load_sd(P + "_target_after_update.npz")
    mine_t = trainer.behaviour_net.target_net.state_dict()
    for k, ref in tgt.items():
        if ref.is_floating_point() and "batchnorm" not in k:
            assert th.allclose(mine_t[k].cpu(), ref, atol=3e-6, rtol=1e-5), k
    assert not util.FALLBACKS, util.FALLBACKS              # no fused path declined the 3-agent shapes


def test_config3_maddpg_3_agents_4096_envs_trains_on_graphs():
    """BASELINE.json config 3 as stated: MADDPG, 3 agents (buildings [5, 15, 25]), 4096 environments on one GPU, device
    replay, graphed rollout and sub-updates (what bench.py's config-3 leg builds).  Two training episodes (190 vector steps:
    three update events) on the product path, then one more value and one more policy sub-update on the SAME replay
    window taken twice — as HIP-graph replays and eagerly from identical weights / optimiser state — must agree, and the
    critic that trained has the reference's shape (maddpg.py:18-27)."""
    import copy
    from safe_marl_amd import util
    from safe_marl_amd.flex_env import VecFlexProvisionEnv
    from safe_marl_amd.learner import MADDPG
    from safe_marl_amd.network import create_network
    from safe_marl_amd.series import make_synthetic_series
    from safe_marl_amd.trainer import PGTrainer
    _loaded_lib()
    util.FALLBACKS.clear()
    n_envs, blds = 4096, [5, 15, 25]
    env_args = {"buildings": blds, "pv_nodes": blds, "ess_nodes": blds}
    net3 = create_network(env_args)
    series = make_synthetic_series(net3, n_days=60)
    env = VecFlexProvisionEnv(env_args, n_envs, net=net3, series=series, seed=1234, warm_start=True)
    d = json.load(open(os.path.join(G, "learner3_args.json")))
    d.update(cuda=True, agent_num=env.n_agents, obs_size=env.obs_size, state_size=env.state_size, v_min=0.9, v_max=1.1)
    th.manual_seed(0)
    np.random.seed(0)
    trainer = PGTrainer(util.convert(d), MADDPG, env, None, batch_scale=n_envs // 4, replay_capacity=n_envs * 96 * 2)
    model = trainer.behaviour_net
    assert model.n_ == 3 and tuple(model.value_dicts[0].fc1.weight.shape) == (64, (144 + 4) * 3 + 3)
    before = {k: v.detach().clone() for k, v in model.state_dict().items()}
    stat = {}
    for _ in range(2):
        model.train_process(stat, trainer)
    th.cuda.synchronize()
    assert trainer.steps == 190
    for k in ("mean_train_reward", "mean_train_value_loss", "mean_train_policy_loss"):
        assert np.isfinite(float(stat[k])), (k, stat[k])
    assert float(stat["mean_train_solver_failed"]) == 0.0 if "mean_train_solver_failed" in stat else True
    moved = [k for k, v in model.state_dict().items() if v.is_floating_point() and not th.equal(v, before[k])]
    assert any(k.startswith("value_dicts.") for k in moved) and any(k.startswith("policy_dicts.") for k in moved)
    assert any(k.startswith("target_net.") for k in moved)                                    # soft update at step 120
    assert sorted(trainer._update_graphs) == ["policy", "value"]                             # graphed sub-updates ran
    assert model._rollout_graph.graph is not None                                            # graphed rollout ran
    assert not util.FALLBACKS, util.FALLBACKS
    # graph replay == eager launches, from the same state on the same window
    snap = copy.deepcopy(model.state_dict())
    osnap = (copy.deepcopy(trainer.value_optimizer.state_dict()), copy.deepcopy(trainer.policy_optimizer.state_dict()))
    rs = np.random.get_state()
    s1 = {}
    trainer.value_replay_process(s1)
    trainer.policy_replay_process(s1)
    th.cuda.synchronize()
    graphed = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model.load_state_dict(snap)
    trainer.value_optimizer.load_state_dict(osnap[0])
    trainer.policy_optimizer.load_state_dict(osnap[1])
    np.random.set_state(rs)
    trainer.graph_updates = False
    s2 = {}
    trainer.value_replay_process(s2)
    trainer.policy_replay_process(s2)
    th.cuda.synchronize()
    for k, v in model.state_dict().items():
        if v.is_floating_point():
            assert th.allclose(v, graphed[k], atol=1e-6, rtol=1e-5), k
    for k in ("mean_train_value_loss", "mean_train_policy_loss", "mean_train_value_grad_norm", "mean_train_policy_grad_norm"):
        assert abs(float(s1[k]) - float(s2[k])) <= 1e-5 * max(1.0, abs(float(s2[k]))), (k, float(s1[k]), float(s2[k]))
