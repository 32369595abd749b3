"""A sub-update replayed as a HIP graph (trainer._graphed_sub_update) equals the eager sub-update on the same batch, and
capturing the graph leaves weights and optimiser state untouched."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _trainer(graph_updates, n_envs=256, alg="maddpg"):
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
    from train_maddpg import DEFAULT_ALG_ARGS
    from safe_marl_amd.flex_env import VecFlexProvisionEnv
    from safe_marl_amd import learner
    from safe_marl_amd.network import create_network
    from safe_marl_amd.series import make_synthetic_series
    from safe_marl_amd.trainer import PGTrainer
    from safe_marl_amd.util import convert
    net = create_network()
    series = make_synthetic_series(net, n_days=30)
    cls = {"maddpg": learner.MADDPG, "matd3": learner.MATD3, "iddpg": learner.IDDPG}[alg]
    name, alg = alg, dict(DEFAULT_ALG_ARGS)
    alg.update(alg=name, agent_num=5, obs_size=144, state_size=110, action_dim=4, behaviour_update_freq=10 ** 9,
               target_update_freq=10 ** 9)
    torch.manual_seed(7)
    np.random.seed(7)
    env = VecFlexProvisionEnv({}, n_envs, net=net, series=series, seed=3, warm_start=True)
    tr = PGTrainer(convert(alg), cls, env, None, replay_capacity=n_envs * 96 * 2, graph_updates=graph_updates)
    tr.behaviour_net.train_process({}, tr)                       # fill the replay (no updates: huge update period)
    return tr


@pytest.mark.parametrize("n_envs", [256, 4096])      # 4096: the update batch of the headline configuration (32 768 samples)
def test_graphed_sub_updates_equal_eager_ones(n_envs):
    a, b = _trainer(True, n_envs), _trainer(False, n_envs)
    for (ka, va), (kb, vb) in zip(a.behaviour_net.state_dict().items(), b.behaviour_net.state_dict().items()):
        assert torch.equal(va, vb), ka                           # same start
    for ring in (a.replay_buffer.obs_source_ring, "hid_ring", "small_ring"):
        assert torch.equal(getattr(a.replay_buffer, ring), getattr(b.replay_buffer, ring)), ring
    for which in ("value", "value", "policy", "policy", "value", "policy"):
        stats = []
        for k, tr in enumerate((a, b)):
            np.random.seed(11 + 3 * len(which))                  # the same replay window for both, another one per kind
            st = {}
            (tr.value_replay_process if which == "value" else tr.policy_replay_process)(st)
            torch.cuda.synchronize()
            stats.append({key: float(v) for key, v in st.items()})
        # the reported statistics are graph outputs too: they must be this replay's values, not stale ones
        assert stats[0].keys() == stats[1].keys()
        for key in stats[0]:
            assert abs(stats[0][key] - stats[1][key]) <= 1e-4 * max(1.0, abs(stats[1][key])), (which, key, stats)
        assert a.graph_updates and which in a._update_graphs     # the graph path really ran
        for (ka, va), (kb, vb) in zip(a.behaviour_net.state_dict().items(), b.behaviour_net.state_dict().items()):
            if va.dtype.is_floating_point:
                assert (va - vb).abs().max().item() <= 2e-6 + 2e-4 * vb.abs().max().item(), (which, ka)


def test_graph_and_eager_stay_bit_identical_over_a_long_schedule():
    """Six update events of the headline configuration's shape (ten value sub-updates, then a policy one) at 4096 envs:
    every kernel on the gradient path reduces in a fixed order and nothing in the graphs depends on replay-time state,
    so weights, optimiser-visible buffers and reported statistics agree to the bit after each of the 66 sub-updates."""
    a, b = _trainer(True, 4096), _trainer(False, 4096)
    for i, which in enumerate((["value"] * 10 + ["policy"]) * 6):
        stats = []
        for tr in (a, b):
            np.random.seed(100 + i)
            st = {}
            (tr.value_replay_process if which == "value" else tr.policy_replay_process)(st)
            torch.cuda.synchronize()
            stats.append({k: float(v) for k, v in st.items()})
        assert stats[0] == stats[1], (i, which, stats)
        for (ka, va), (kb, vb) in zip(a.behaviour_net.state_dict().items(), b.behaviour_net.state_dict().items()):
            assert torch.equal(va, vb), (i, which, ka)


def test_graphed_regions_contain_no_aten_multiblock_reduction(monkeypatch):
    """Guard rail for DESIGN.md §6: with FLEX_GRAPH_AUDIT=1 every capture site first runs its body under torch.profiler
    and refuses kernels of util.GRAPH_DENYLIST (ATen reduce_kernel / batch_norm statistics).  Both sub-update graphs
    and the fused rollout graph of MADDPG pass, and the audit really saw this project's kernels."""
    monkeypatch.setenv("FLEX_GRAPH_AUDIT", "1")
    tr = _trainer(True, 256)
    rg = tr.behaviour_net._rollout_graph
    for which in ("value", "policy"):
        np.random.seed(3)
        (tr.value_replay_process if which == "value" else tr.policy_replay_process)({})
    torch.cuda.synchronize()
    assert tr.graph_updates and set(tr._update_graphs) == {"value", "policy"}
    for which, must in (("value", ("td_", "critic_tail", "clip_rmsprop", "actor_")),
                        ("policy", ("gru_backward_fused", "wgrad", "clip_rmsprop", "sum_partial"))):
        names = tr.graph_audit[which]
        for m in must:
            assert any(m in k for k in names), (which, m, names)
        assert not any("at::native::reduce_kernel" in k or "batch_norm" in k for k in names), names
    # the fused rollout step is TWO kernels: the policy and the env step that files the transition itself
    # (actor_rollout16_kernel at rollout sizes, actor_forward_mfma_kernel above 80 rows per CU)
    assert rg.sink_active and any("actor_" in k for k in rg.audit) and any("flex_step_kernel" in k for k in rg.audit)
    assert not any("rollout_pack_kernel" in k for k in rg.audit)
    # and the guard itself fires on an ATen full reduction
    from safe_marl_amd.util import audit_graph_body
    x = torch.randn(1 << 20, device="cuda")
    with pytest.raises(RuntimeError, match="multi-block"):
        audit_graph_body(lambda: x.sum())


def test_fused_paths_do_not_fall_back_at_the_default_configuration():
    """Guard rail (c): a training episode + both sub-updates at the default configuration take no PyTorch fallback of a
    fused path (util.FALLBACKS stays empty); a drifted configuration is reported once."""
    import warnings
    from safe_marl_amd import util
    util.FALLBACKS.clear()
    tr = _trainer(True, 256)
    for which in ("value", "policy"):
        np.random.seed(3)
        (tr.value_replay_process if which == "value" else tr.policy_replay_process)({})
    torch.cuda.synchronize()
    assert util.FALLBACKS == {}, util.FALLBACKS
    from safe_marl_amd.nets import fused_actor_forward
    agent = tr.behaviour_net.policy_dicts[0]
    wide = torch.zeros(4, 5, 200, device="cuda")            # 200 observation columns: outside csrc/actor.hip
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert fused_actor_forward(agent, wide, torch.zeros(20, 64, device="cuda"), 5, True) is None
        assert fused_actor_forward(agent, wide, torch.zeros(20, 64, device="cuda"), 5, True) is None
    assert util.FALLBACKS.get("actor_forward") == 2 and sum("actor_forward" in str(x.message) for x in w) == 1
    util.FALLBACKS.clear()


def test_pipelined_update_event_equals_one_at_a_time_sub_updates():
    """trainer.replay_event (model.py:47-50: ten value sub-updates, then a policy one) with the next window gathered on a
    side stream into the second static batch while the current graph runs, against the same event done one call at a
    time: same windows (same NumPy stream), same graphs — weights, optimiser-visible buffers and statistics bit-identical
    over three events at the headline batch."""
    a, b = _trainer(True, 4096), _trainer(True, 4096)
    a.pipeline_updates, b.pipeline_updates = True, False
    for ev in range(3):
        stats = []
        for tr in (a, b):
            np.random.seed(40 + ev)
            st = {}
            tr.replay_event(st, 10, 1)
            torch.cuda.synchronize()
            stats.append({k: float(v) for k, v in st.items()})
        assert stats[0] == stats[1], (ev, stats)
        for (ka, va), (kb, vb) in zip(a.behaviour_net.state_dict().items(), b.behaviour_net.state_dict().items()):
            assert torch.equal(va, vb), (ev, ka)
    # the double buffer really was used (of the plain value sub-update, or of its form on filed bootstrap values)
    assert (set(a._update_graphs_alt) == {"value"} or ("value_cached", 1) in a._cached_graphs) and not b._update_graphs_alt
    # ... by the side-stream pipeline in `a` (no event graph there); `b` ran its events as one graph each (round 5)
    assert a.event_graph_replays == 0 and b.event_graph_replays == 3


@pytest.mark.parametrize("alg", ["matd3", "iddpg"])
def test_matd3_and_iddpg_sub_updates_replay_as_graphs(alg, monkeypatch):
    """Round 2: the GPU path of MATD3's and IDDPG's losses reduces only through this project's kernels (twin critic nodes,
    flexnet_td_loss, pointwise agent sums), so their sub-updates are captured like MADDPG's.  The capture-time audit
    accepts both bodies; the policy sub-update (no random draw inside) replays bit-identically to the eager step; the value
    sub-update (MATD3 draws its target-smoothing noise inside, matd3.py:136-138) moves the critic and stays finite."""
    monkeypatch.setenv("FLEX_GRAPH_AUDIT", "1")
    a, b = _trainer(True, 256, alg), _trainer(False, 256, alg)
    for ka, kb in zip(a.behaviour_net.state_dict().values(), b.behaviour_net.state_dict().values()):
        assert torch.equal(ka, kb)
    for which in ("policy", "policy"):
        stats = []
        for tr in (a, b):
            np.random.seed(5)
            st = {}
            tr.policy_replay_process(st)
            torch.cuda.synchronize()
            stats.append({k: float(v) for k, v in st.items()})
        assert stats[0] == stats[1], stats
        for (ka, va), (kb, vb) in zip(a.behaviour_net.state_dict().items(), b.behaviour_net.state_dict().items()):
            assert torch.equal(va, vb), (alg, ka)
    before = {k: v.clone() for k, v in a.behaviour_net.value_dicts.state_dict().items()}
    for _ in range(3):
        st = {}
        np.random.seed(6)
        a.value_replay_process(st)
    torch.cuda.synchronize()
    assert a.graph_updates and set(a._update_graphs) == {"value", "policy"} and not b._update_graphs
    assert np.isfinite(float(st["mean_train_value_loss"])) and np.isfinite(float(st["mean_train_value_grad_norm"]))
    after = a.behaviour_net.value_dicts.state_dict()
    assert all(torch.isfinite(v).all() for v in after.values()) and any(not torch.equal(before[k], after[k]) for k in before)
    for which in ("value", "policy"):
        names = a.graph_audit[which]
        assert not any("at::native::reduce_kernel" in k or "batch_norm" in k for k in names), (alg, which, names)
        assert any("clip_rmsprop" in k for k in names)


def test_value_steps_with_the_td_error_formed_in_the_backward_track_the_sequence():
    """Twelve value sub-updates at the headline batch (32 768 samples x 5 agents): nets._CriticTdLossFn (the TD error
    formed inside the critic's backward kernel) against the forward -> flexnet_td_loss -> backward sequence from the same
    start on the same replay windows.  Different summation orders, same mathematics: the reported losses agree to 1e-5
    relative, parameters and RMSprop state stay within fp32 round-off of each other, BatchNorm statistics likewise."""
    from safe_marl_amd import learner
    a, b = _trainer(True, 4096), _trainer(True, 4096)
    b.behaviour_net.fused_td_backward = False                  # instance attribute: this trainer's learner only
    for i in range(12):
        stats = []
        for tr in (a, b):
            np.random.seed(300 + i)
            st = {}
            tr.value_replay_process(st)
            torch.cuda.synchronize()
            stats.append(float(st["mean_train_value_loss"]))
        assert abs(stats[0] - stats[1]) <= 1e-5 * abs(stats[1]), (i, stats)
    na = [k for k in a.graph_audit["value"]] if getattr(a, "graph_audit", None) else None
    for (ka, va), (kb, vb) in zip(a.behaviour_net.state_dict().items(), b.behaviour_net.state_dict().items()):
        if va.dtype.is_floating_point:
            # (3e-6: the post-RMSprop tolerance of the golden tests — twelve steps of lr 1e-4 on gradients that are sums over
            # 163 840 rows in two summation orders; LayerNorm's bias starts at zero, so the relative term is no help there)
            assert (va - vb).abs().max().item() <= 3e-6 + 1e-4 * vb.abs().max().item(), ka
    oa, ob = a.value_optimizer.state_dict()["state"], b.value_optimizer.state_dict()["state"]
    for k in oa:
        # (relative to the tensor's largest entry: an entry whose gradient is a near-cancelling sum over 163 840 rows moves by
        # parts in 10^3 with the order the rows are summed in, and the two paths round dLoss/dq differently)
        sa, sb = oa[k]["square_avg"], ob[k]["square_avg"]
        assert (sa - sb).abs().max().item() <= 1e-3 * sb.abs().max().item() + 1e-12, k
    assert type(a.behaviour_net)._critic_td_loss is learner._maddpg_critic_td_loss


@pytest.mark.parametrize("pipelined,episodes,alg", [(False, 1, "maddpg"), (True, 1, "maddpg"), (False, 3, "maddpg"),
                                                    (False, 1, "iddpg")])
def test_bootstrap_values_filed_once_per_event_change_nothing(pipelined, episodes, alg):
    """Round 3, trainer.replay_event at the reference's sample reuse (batch = 32 x n_envs transitions = 32 slabs of a ring
    that holds 95): the windows of the ten value sub-updates are drawn up front, the union of their transitions gets its
    Q'(s', pi(s')) in a few passes of one batch (a graph of MADDPG.bootstrap_values) into the replay's nv_ring, and the
    value sub-updates read them from there — against the same trainer computing them inside every sub-update: same windows,
    same kernels on the same rows: weights, optimiser state and statistics bit-identical over three events (also with the ring
    wrapped, where passes and windows lie astride its seam); and the cached
    form really ran (fewer passes than sub-updates), while at the default batch (8 slabs per window) it does not."""
    from safe_marl_amd.trainer import PGTrainer
    # [0, 30) and [100, 125): a whole pass each and one flush right
    assert PGTrainer.bootstrap_chunks([0, 10, 100, 105], 20) == [(0, 20), (10, 20), (100, 20), (105, 20)]
    # [0, 90) and [95, 135): three passes + one — joined across the gap [90, 95) (same run of transitions): 135 / 40 -> four
    assert PGTrainer.bootstrap_chunks([50, 0, 10, 95], 40) == [(0, 40), (40, 40), (50, 40), (95, 40)]
    assert PGTrainer.bootstrap_chunks([50, 0, 10, 80], 40, [(0, 500)]) == [(0, 40), (40, 40), (80, 40)]
    assert PGTrainer.bootstrap_chunks([0, 45, 100], 40, [(0, 500)]) == [(0, 40), (45, 40), (100, 40)]      # joining: 4 > 3
    # [0, 45) and [50, 95): two passes each, three joined — but not across two runs of transitions
    assert PGTrainer.bootstrap_chunks([0, 5, 50, 55], 40, [(0, 500)]) == [(0, 40), (40, 40), (55, 40)]
    assert PGTrainer.bootstrap_chunks([0, 5, 50, 55], 40, [(0, 48), (48, 500)]) == [(0, 40), (5, 40), (50, 40), (55, 40)]
    a, b = _trainer(True, 1024, alg), _trainer(True, 1024, alg)
    for tr in (a, b):
        for _ in range(episodes - 1):                 # three episodes = 285 slabs into a ring of 192: windows astride its seam
            tr.behaviour_net.train_process({}, tr)
        tr.batch_scale = 1024
        tr.pipeline_updates = pipelined
    b.cache_bootstrap = False
    assert (a.replay_buffer.k > a.replay_buffer.slabs) == (episodes == 3)
    for ev in range(3):
        stats = []
        for tr in (a, b):
            np.random.seed(70 + ev)
            st = {}
            tr.replay_event(st, 10, 1)
            torch.cuda.synchronize()
            stats.append({k: float(v) for k, v in st.items()})
        assert stats[0] == stats[1], (ev, stats)
        for (ka, va), (kb, vb) in zip(a.behaviour_net.state_dict().items(), b.behaviour_net.state_dict().items()):
            assert torch.equal(va, vb), (ev, ka)
        for pa, pb in zip(a.value_optimizer.param_groups[0]["params"], b.value_optimizer.param_groups[0]["params"]):
            assert torch.equal(a.value_optimizer.state[pa]["square_avg"], b.value_optimizer.state[pb]["square_avg"])
    assert a.bootstrap_cached_events == 3 and b.bootstrap_cached_events == 0
    assert sorted(a._update_graphs) == ["policy", "value"]                       # (the cached form is kept apart)
    c = _trainer(True, 1024)                                                     # default batch: windows hardly overlap
    np.random.seed(1)
    c.replay_event({}, 10, 1)
    torch.cuda.synchronize()
    assert c.bootstrap_cached_events == 0
    # MATD3's value loss has its own get_loss (min of twins, target-smoothing noise drawn inside): never on filed values
    m = _trainer(True, 1024, "matd3")
    m.batch_scale = 1024
    m.replay_event({}, 10, 1)
    torch.cuda.synchronize()
    assert m.bootstrap_cached_events == 0 and not m._cached_graphs


@pytest.mark.parametrize("which", ["value", "value_cached"])
def test_a_failed_capture_leaves_weights_optimiser_state_and_statistics_untouched(which, monkeypatch):
    """the warm-up steps a capture needs are REAL optimiser steps; whatever ends the capture — here an exception
    injected into the captured region itself, after the warm-up has run — they are undone (try / finally), so the eager
    fallback ("value") and the fall-back to per-sub-update bootstrap values ("value_cached") start from the weights, RMSprop
    state and reward-BatchNorm running statistics the schedule of model.py:43-50 had reached."""
    import warnings
    from safe_marl_amd import util
    tr = _trainer(True, 256)
    if which == "value_cached":
        tr.batch_scale = 256
    net = tr.behaviour_net
    # (the optimiser state exists already: one eager value step)
    np.random.seed(5)
    tr._sub_update("value", {}, tr.replay_buffer.get_batch_tensors(tr.effective_batch_size()))
    before = {k: v.detach().clone() for k, v in net.state_dict().items()}
    opt_before = [tr.value_optimizer.state[p]["square_avg"].clone() for p in tr.value_optimizer.param_groups[0]["params"]]
    real, calls = util.graph_capture.__enter__, []

    def failing_enter(self):
        calls.append(1)
        raise RuntimeError("injected: capture refused")

    monkeypatch.setattr(util.graph_capture, "__enter__", failing_enter)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        g = tr._ensure_graph(which)
    monkeypatch.setattr(util.graph_capture, "__enter__", real)
    torch.cuda.synchronize()
    assert g is None and calls and any("capture" in str(x.message) for x in w)
    assert tr.graph_updates == (which == "value_cached") and tr.cache_bootstrap == (which != "value_cached")
    for k, v in net.state_dict().items():
        assert torch.equal(v, before[k]), k                     # incl. batchnorm.running_mean / running_var / num_batches_tracked
    for p, sq in zip(tr.value_optimizer.param_groups[0]["params"], opt_before):
        assert torch.equal(tr.value_optimizer.state[p]["square_avg"], sq)
    assert net.bootstrap_from_batch is False
    # and the trainer still trains: the next sub-update runs (eagerly, or as a graph on freshly computed bootstrap values)
    st = {}
    tr.value_replay_process(st)
    torch.cuda.synchronize()
    assert np.isfinite(float(st["mean_train_value_loss"]))
    assert any(not torch.equal(v, before[k]) for k, v in net.state_dict().items() if k.startswith("value_dicts."))


def test_value_sub_updates_read_their_window_in_place_from_the_stacked_ring():
    """Round 5: with the bootstrap values filed per event, the value sub-update's observations are read IN
    PLACE from the replay's stacked-observation ring (replay_buffer.enable_stacked_ring: every slab expanded once; the first
    layer's flexnet_linear2 and its flexnet_wgrad take the ring's base and a device cell with the window's first row) instead
    of being gathered into a static batch per sub-update.  Same kernels on the same values: three events — the ring wrapped,
    windows astride its seam — leave weights, optimiser state and statistics bit-identical to the gathering form
    (FLEX_STACKED_RING=0), and the in-place form really ran: a NaN placeholder stands where the gathered copy used to be, and
    no gather_window launch is issued for the value sub-updates."""
    import os
    from safe_marl_amd import nets
    a = _trainer(True, 1024)
    os.environ["FLEX_STACKED_RING"] = "0"
    try:
        b = _trainer(True, 1024)
        for tr in (a, b):
            for _ in range(2):
                tr.behaviour_net.train_process({}, tr)
            tr.batch_scale = 1024
        for ev in range(3):
            stats = []
            for tr, flag in ((a, "1"), (b, "0")):
                os.environ["FLEX_STACKED_RING"] = flag
                np.random.seed(90 + ev)
                st = {}
                tr.replay_event(st, 10, 1)
                torch.cuda.synchronize()
                stats.append({k: float(v) for k, v in st.items()})
            assert stats[0] == stats[1], (ev, stats)
            for (ka, va), (kb, vb) in zip(a.behaviour_net.state_dict().items(), b.behaviour_net.state_dict().items()):
                assert torch.equal(va, vb), (ev, ka)
    finally:
        os.environ.pop("FLEX_STACKED_RING", None)
    assert a.bootstrap_cached_events == 3 and b.bootstrap_cached_events == 3
    ga, gb = a._cached_graphs[("value_cached", 0)], b._cached_graphs[("value_cached", 0)]
    assert [p[0] for p in ga["plan"]].count("stack_ring") == 1 and "row_ring" not in [p[0] for p in ga["plan"]]
    assert "row_ring" in [p[0] for p in gb["plan"]] and "stack_ring" not in [p[0] for p in gb["plan"]]
    assert torch.isnan(ga["batch"].state).all()                       # the placeholder, never written
    # ... and so do the passes that file the bootstrap values: policy and target critic read next_state in place
    for g in a._bootstrap_graphs.values():
        assert [p[0] for p in g["plan"]].count("stack_ring") == 1 and "row_ring" not in [p[0] for p in g["plan"]]
        assert torch.isnan(g["batch"].next_state).all()
        assert torch.isnan(g["batch"].hid).all()                      # the hidden states too (mirrored tail of the hidden-state ring)
    for g in b._bootstrap_graphs.values():
        assert "row_ring" in [p[0] for p in g["plan"]]
    assert a.replay_buffer.stack_ring is not None and getattr(b.replay_buffer, "stack_ring", None) is None
    # the ring holds what a gather of the same window forms
    buf = a.replay_buffer
    buf.expand_stacked()
    slot = buf.sample_slot(4096)
    want = buf.stacked_obs(slot, 4096)
    p = slot % buf.stack_rows
    assert torch.equal(buf.stack_ring[p:p + 4096], want)
    assert nets.ring_view_of(ga["batch"].state.reshape(ga["bs"], -1)) is not None


def test_plain_value_sub_updates_read_both_views_of_their_window_in_place():
    """The value sub-update that computes its own bootstrap values (the default batch: windows hardly overlap, nothing is filed
    per event) reads `state` and `next_state` — N ring rows apart — in place as well: policy inference and target critic on
    next_state, behaviour critic's first layer and its weight gradient on state.  Five sub-updates and a policy sub-update
    against the gathering form (FLEX_STACKED_RING=0): bit-identical weights, optimiser state, statistics."""
    import os
    a = _trainer(True, 4096)
    os.environ["FLEX_STACKED_RING"] = "0"
    try:
        b = _trainer(True, 4096)
        for i in range(5):
            stats = []
            for tr, flag in ((a, "1"), (b, "0")):
                os.environ["FLEX_STACKED_RING"] = flag
                np.random.seed(40 + i)
                st = {}
                tr.value_replay_process(st)
                if i == 4:
                    tr.policy_replay_process(st)
                torch.cuda.synchronize()
                stats.append({k: float(v) for k, v in st.items()})
            assert stats[0] == stats[1], (i, stats)
    finally:
        os.environ.pop("FLEX_STACKED_RING", None)
    for (ka, va), (kb, vb) in zip(a.behaviour_net.state_dict().items(), b.behaviour_net.state_dict().items()):
        assert torch.equal(va, vb), ka
    pa, pb = [p[0] for p in a._update_graphs["value"]["plan"]], [p[0] for p in b._update_graphs["value"]["plan"]]
    assert pa.count("stack_ring") == 1 and "row_ring" not in pa and "row_ring" in pb
    assert torch.isnan(a._update_graphs["value"]["batch"].state).all() and torch.isnan(a._update_graphs["value"]["batch"].next_state).all()
    assert torch.isnan(a._update_graphs["value"]["batch"].hid).all()
    assert "hid_ring" not in pa and "hid_ring" in pb


@pytest.mark.parametrize("reuse", [False, True])
def test_an_events_value_sub_updates_as_one_graph_change_nothing(reuse, monkeypatch):
    """Round 5: trainer.replay_event runs the ten value sub-updates of an update event (model.py:47-50) as
    ONE HIP graph — every sub-update refreshing its static batch from its own cell of a device array of window starts
    (flexnet_window_refresh), the refresh for the next one riding in the optimiser step's launches
    (flexnet_clip_rmsprop_refresh), the reward statistics riding in the refresh — against one graph launch per sub-update with
    the refresh issued from the host (trainer.event_graphs off): the same kernels on the same windows in the same order, so the
    same bits, at the default batch (plain value sub-updates) and at the reference's sample reuse (bootstrap values filed per
    event, "value_cached")."""
    n_envs = 1024 if reuse else 4096
    a, b = _trainer(True, n_envs), _trainer(True, n_envs)
    b.event_graphs = False
    if reuse:
        for tr in (a, b):
            tr.batch_scale = 1024                                # 32 slabs per window of a ring that holds 95: the windows overlap
    for ev in range(3):
        stats = []
        for tr in (a, b):
            np.random.seed(500 + ev)
            st = {}
            tr.replay_event(st, 10, 1)
            torch.cuda.synchronize()
            stats.append({k: float(v) for k, v in st.items()})
        assert stats[0] == stats[1], (ev, stats)
        for (ka, va), (kb, vb) in zip(a.behaviour_net.state_dict().items(), b.behaviour_net.state_dict().items()):
            assert torch.equal(va, vb), (ev, ka)
    assert a.event_graph_replays == 3 and b.event_graph_replays == 0
    kind = "value_cached" if reuse else "value"
    assert a.bootstrap_cached_events == (3 if reuse else 0)
    eg = a._event_graphs[(kind, 10)]
    assert eg and eg["base"].get("td") is not None              # the statistics ride in the refresh
