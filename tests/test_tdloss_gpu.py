"""csrc/tdloss.hip (include/flexnet.h: flexnet_td_loss) — the value loss of maddpg.py:100-123 behind the reward
normalisation of model.py:308-323, its gradient, and the BatchNorm running statistics — against PyTorch."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _torch_loss(q, next_q, reward, done, gamma, bn):
    r = reward
    if bn is not None:
        with torch.no_grad():
            r = bn(reward)
    ret = r + gamma * (1 - done.view(-1, 1)) * next_q
    return (ret - q).pow(2).mean()


@pytest.mark.parametrize("rows,n,norm", [(32768, 5, True), (4099, 3, True), (2, 8, True), (1000, 5, False), (1, 1, False)])
def test_loss_gradient_and_running_statistics(rows, n, norm):
    from safe_marl_amd.nets import td_loss, td_loss_supported
    g = torch.Generator(device="cuda").manual_seed(rows + n)
    reward = torch.randn(rows, n, device="cuda", generator=g) * 3.0 - 5.0
    done = (torch.rand(rows, device="cuda", generator=g) < 0.1).float()
    next_q = torch.randn(rows, n, device="cuda", generator=g)
    q0 = torch.randn(rows, n, device="cuda", generator=g)
    bns = []
    for _ in range(2):
        bn = None
        if norm:
            bn = torch.nn.BatchNorm1d(n).cuda()
            with torch.no_grad():
                bn.weight.copy_(torch.linspace(0.5, 1.5, n)); bn.bias.copy_(torch.linspace(-0.2, 0.2, n))
                bn.running_mean.fill_(0.3); bn.running_var.fill_(2.0)
        bns.append(bn)
    qa, qb = q0.clone().requires_grad_(True), q0.clone().requires_grad_(True)
    assert td_loss_supported(qa, next_q, reward, done, bns[0])
    la = td_loss(qa, next_q, reward, done, 0.99, bns[0])
    lb = _torch_loss(qb, next_q, reward, done, 0.99, bns[1])
    ga, = torch.autograd.grad(la * 1.7, [qa])                  # an upstream factor, as a scaled loss would give
    gb, = torch.autograd.grad(lb * 1.7, [qb])
    assert abs(la.item() - lb.item()) <= 2e-5 * max(1.0, abs(lb.item()))
    assert (ga - gb).abs().max().item() <= 2e-5 * max(1e-9, gb.abs().max().item())
    if norm:
        assert torch.allclose(bns[0].running_mean, bns[1].running_mean, rtol=1e-5, atol=1e-6)
        assert torch.allclose(bns[0].running_var, bns[1].running_var, rtol=1e-5, atol=1e-6)
        assert bns[0].num_batches_tracked.item() == bns[1].num_batches_tracked.item() == 1


def test_get_loss_value_branch_matches_the_pytorch_composition():
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
    from train_maddpg import DEFAULT_ALG_ARGS
    from safe_marl_amd import learner
    from safe_marl_amd.replay_buffer import Transition
    from safe_marl_amd.util import convert
    alg = dict(DEFAULT_ALG_ARGS)
    alg.update(alg="maddpg", agent_num=5, obs_size=144, state_size=110, action_dim=4)
    b = 4096
    g = torch.Generator(device="cuda").manual_seed(12)
    fields = dict(state=torch.randn(b, 5, 144, device="cuda", generator=g), action=torch.rand(b, 5, 4, device="cuda", generator=g),
                  log_prob_a=torch.zeros(b, 5, 4, device="cuda"), value=torch.zeros(b, 5, 1, device="cuda"),
                  next_value=torch.zeros(b, 5, 1, device="cuda"), reward=torch.randn(b, 5, device="cuda", generator=g),
                  next_state=torch.randn(b, 5, 144, device="cuda", generator=g),
                  done=(torch.rand(b, device="cuda", generator=g) < 0.05).float(), last_step=torch.zeros(b, device="cuda"),
                  action_avail=torch.ones(b, 5, 4, device="cuda"), last_hid=torch.randn(b, 5, 64, device="cuda", generator=g),
                  hid=torch.randn(b, 5, 64, device="cuda", generator=g))
    batch = Transition(**fields)
    out = []
    for fused in (True, False):
        torch.manual_seed(5)
        m = learner.MADDPG(convert(alg), learner.MADDPG(convert(alg)).cuda()).cuda()
        with torch.no_grad():
            for p in m.parameters():
                p.copy_(torch.randn_like(p) * 0.1)
        saved = learner.td_loss_supported
        if not fused:
            learner.td_loss_supported = lambda *a, **k: False
        try:
            _, vloss, _ = m.get_loss(batch, need="value")
            grads = torch.autograd.grad(vloss, list(m.value_dicts.parameters()))
        finally:
            learner.td_loss_supported = saved
        out.append((vloss.detach(), grads, m.batchnorm.running_mean.clone(), m.batchnorm.running_var.clone(),
                    int(m.batchnorm.num_batches_tracked)))
    assert abs(out[0][0].item() - out[1][0].item()) <= 1e-5 * max(1.0, abs(out[1][0].item()))
    for a, e in zip(out[0][1], out[1][1]):
        assert (a - e).abs().max().item() <= 1e-4 * max(1e-6, e.abs().max().item())
    assert torch.allclose(out[0][2], out[1][2], rtol=1e-5, atol=1e-6) and torch.allclose(out[0][3], out[1][3], rtol=1e-5, atol=1e-6)
    assert out[0][4] == out[1][4] == 1


def test_statistics_only_pass_moves_the_running_statistics_like_a_forward():
    from safe_marl_amd.nets import batchnorm_stats_supported, batchnorm_update_running_stats
    g = torch.Generator(device="cuda").manual_seed(21)
    a, b = torch.nn.BatchNorm1d(5).cuda(), torch.nn.BatchNorm1d(5).cuda()
    for step in range(3):
        x = torch.randn(32768, 5, device="cuda", generator=g) * (1 + step) + step
        assert batchnorm_stats_supported(a, x)
        batchnorm_update_running_stats(a, x)
        with torch.no_grad():
            b(x)
        assert torch.allclose(a.running_mean, b.running_mean, rtol=1e-5, atol=1e-6)
        assert torch.allclose(a.running_var, b.running_var, rtol=1e-5, atol=1e-6)
        assert a.num_batches_tracked.item() == b.num_batches_tracked.item() == step + 1
    a.eval()
    assert not batchnorm_stats_supported(a, x)


def test_unit_seed_is_the_gradient_of_one_without_the_launches():
    """util.unit_seed: the trainer's root gradient.  The loss nodes recognise it by address and hand back their stored /
    constant gradient (no ones-fill, no multiply by one); any other root gradient takes the general path.  Same numbers
    either way, and the seed itself is never written."""
    import torch.nn as nn
    from safe_marl_amd.nets import td_loss
    from safe_marl_amd.util import mean_all, unit_seed, is_unit_seed, audit_graph_body
    g = torch.Generator(device="cuda").manual_seed(1)
    rows, n = 4096, 5
    q = torch.randn(rows, n, device="cuda", generator=g, requires_grad=True)
    nq, r = torch.randn(rows, n, device="cuda", generator=g), torch.randn(rows, n, device="cuda", generator=g)
    d = (torch.rand(rows, 1, device="cuda", generator=g) < 0.1).float()
    seed = unit_seed("cuda")
    assert is_unit_seed(seed) and not is_unit_seed(torch.ones((), device="cuda")) and seed is unit_seed(q.device)

    def grads(root):
        bn = nn.BatchNorm1d(n).cuda().train()
        loss = td_loss(q, nq, r, d, 0.99, bn)
        pol = mean_all(q * 1.0, sign=-1.0) - 0.25
        return torch.autograd.grad(loss, q, grad_outputs=root)[0], torch.autograd.grad(pol, q, grad_outputs=root)[0]

    a_v, a_p = grads(seed)
    b_v, b_p = grads(None)
    c_v, c_p = grads(torch.full((), 2.0, device="cuda"))
    assert torch.equal(a_v, b_v) and torch.equal(a_p, b_p)
    assert torch.equal(c_v, 2.0 * b_v) and torch.allclose(c_p, 2.0 * b_p, rtol=1e-7, atol=0)
    assert float(seed) == 1.0
    # and the launches are really gone: the value-loss backward under the seed is free of pointwise kernels
    bn = nn.BatchNorm1d(n).cuda().train()
    loss = td_loss(q, nq, r, d, 0.99, bn)
    names = audit_graph_body(lambda: torch.autograd.grad(loss, q, grad_outputs=seed))
    assert not [k for k in names if "elementwise" in k], names


def test_cross_rank_statistics_path_with_two_identical_ranks(monkeypatch):
    """include/flexnet.h stats_ready / stat_rows (trainer.sync_reward_bn): flexnet_td_stats, an all-reduce of the per-block
    partial sums, then the loss call on the summed statistics.  Emulated on one GPU with a stand-in all-reduce that doubles
    its argument — two ranks holding the same shard: the batch mean and the biased variance are those of one shard (sums
    and row count both double: exact in fp64), so loss and gradient equal the single-rank call BIT FOR BIT, and the running
    variance moves with the unbiased factor of 2 rows instead of rows.  Same for the critic's fused TD backward."""
    import torch.distributed as dist
    from safe_marl_amd import nets
    from safe_marl_amd.nets import td_loss
    rows, n = 32768, 5
    g = torch.Generator(device="cuda").manual_seed(77)
    reward = torch.randn(rows, n, device="cuda", generator=g) * 2.0 + 1.0
    done = (torch.rand(rows, device="cuda", generator=g) < 0.1).float()
    next_q = torch.randn(rows, n, device="cuda", generator=g)
    q0 = torch.randn(rows, n, device="cuda", generator=g)

    def run(sync):
        bn = torch.nn.BatchNorm1d(n).cuda()
        q = q0.clone().requires_grad_(True)
        if sync:
            monkeypatch.setattr(nets, "_sync_active", lambda bn=None: True)
            monkeypatch.setattr(dist, "all_reduce", lambda t, op=None: t.mul_(2.0))
            monkeypatch.setattr(dist, "get_world_size", lambda *a, **k: 2)
        try:
            loss = td_loss(q, next_q, reward, done, 0.99, bn)
            gq, = torch.autograd.grad(loss, [q])
        finally:
            monkeypatch.undo()
        return loss, gq, bn

    la, ga, bna = run(False)
    lb, gb, bnb = run(True)
    assert torch.equal(la, lb) and torch.equal(ga, gb)
    assert torch.allclose(bna.running_mean, bnb.running_mean, rtol=0, atol=1e-7)
    var = reward.double().var(0, unbiased=False)
    want = 0.9 * 1.0 + 0.1 * var * (2 * rows) / (2 * rows - 1)
    assert torch.allclose(bnb.running_var.double(), want, rtol=1e-6)
    assert not torch.equal(bna.running_var, bnb.running_var) or rows > 10 ** 7


@pytest.mark.parametrize("rows,wrap", [(32768, False), (32768, True), (4099, True)])
def test_statistics_ride_in_the_gather_launch(rows, wrap):
    """include/flexnet.h: flexnet_gather_rows_td — the statistics pass of the value loss (model.py:308-323) as extra blocks of
    the launch that copies the sampled window's small columns out of the replay ring (utils/replay_buffer.py:17-21).  The
    statistics are taken from the rows the copy READS, in flexnet_td_stats' partition: the workspace must hold the same bits
    as flexnet_td_stats run on the copied tensor, for a window in one piece and for one that wraps the ring's seam."""
    import ctypes as C
    from safe_marl_amd import _lib
    lib = _lib.load()
    n, act_w, stride, cap = 5, 20, 27, 40000
    g = torch.Generator(device="cuda").manual_seed(rows)
    ring = torch.randn(cap, stride, device="cuda", generator=g) * 2.0 + 0.5
    first = cap - rows // 3 if wrap else 1234
    segs = [(first, min(rows, cap - first))] + ([(0, rows - (cap - first))] if first + rows > cap else [])
    reward = torch.full((rows, n), float("nan"), device="cuda")
    action = torch.full((rows, act_w), float("nan"), device="cuda")
    a = _lib.FlexGatherArgs()
    jobs, done_rows = [], 0
    for p, c in segs:                                  # the action columns first, then the reward's one or two pieces
        jobs.append((ring.data_ptr() + 4 * p * stride, action.data_ptr() + 4 * done_rows * act_w, c, act_w))
        done_rows += c
    reward_job, done_rows = len(jobs), 0
    for p, c in segs:
        jobs.append((ring.data_ptr() + 4 * (p * stride + act_w), reward.data_ptr() + 4 * done_rows * n, c, n))
        done_rows += c
    for j, (src, dst, c, w) in enumerate(jobs):
        a.src[j], a.dst[j], a.rows[j], a.width[j], a.src_stride[j], a.dst_stride[j] = src, dst, c, w, stride, w
    a.n_jobs = len(jobs)
    ws = [torch.zeros(_lib.FLEXNET_TD_WS_FLOATS // 2, dtype=torch.float64, device="cuda") for _ in range(2)]
    t = _lib.FlexTdLossArgs()
    t.rows, t.n_agents, t.normalise, t.reward = rows, n, 1, reward.data_ptr()
    t.workspace, t.workspace_floats = ws[0].data_ptr(), 2 * ws[0].numel()
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(lib.flexnet_gather_rows_td(C.byref(a), reward_job, len(segs), C.byref(t), stream), "flexnet_gather_rows_td")
    torch.cuda.synchronize()
    idx = (first + torch.arange(rows, device="cuda")) % cap
    assert torch.equal(reward, ring[idx, act_w:act_w + n]) and torch.equal(action, ring[idx, :act_w])
    t.workspace = ws[1].data_ptr()
    _lib.check(lib.flexnet_td_stats(C.byref(t), stream), "flexnet_td_stats")
    torch.cuda.synchronize()
    k = _lib.FLEXNET_TD_STAT_DOUBLES
    assert torch.equal(ws[0][:k], ws[1][:k]) and bool(ws[1][:k].abs().sum() > 0)
    # arguments that do not describe the reward's copies are refused before any launch
    assert lib.flexnet_gather_rows_td(C.byref(a), 0, 1, C.byref(t), stream) != 0            # job 0 copies the action columns
    assert lib.flexnet_gather_rows_td(C.byref(a), reward_job, 3, C.byref(t), stream) != 0
    t.rows = rows - 1
    assert lib.flexnet_gather_rows_td(C.byref(a), reward_job, len(segs), C.byref(t), stream) != 0


@pytest.mark.parametrize("start", [1234, 40000 * 7 + 39000, 40000 * 3 - 1])
def test_window_refresh_from_a_device_cell_equals_the_host_side_gather(start):
    """include/flexnet.h: flexnet_window_refresh — the refresh of a static batch with the window's first slot read from device
    memory (what lets ONE HIP graph hold every value sub-update of an update event, model.py:47-50): copies, first-row cells
    and the riding statistics pass equal flexnet_gather_rows_td's on the same window, seam included, bit for bit."""
    import ctypes as C
    from safe_marl_amd import _lib
    lib = _lib.load()
    rows, n, act_w, stride, cap, row_off = 8192, 5, 20, 27, 40000, 4096
    g = torch.Generator(device="cuda").manual_seed(start % 1000)
    small = torch.randn(cap, stride, device="cuda", generator=g)
    nv = torch.randn(cap, n, device="cuda", generator=g)
    out = {k: torch.full((rows, w), float("nan"), device="cuda") for k, w in (("action", act_w), ("reward", n), ("done", 1), ("nv", n))}
    cell = torch.full((2,), -1, dtype=torch.int64, device="cuda")
    start_cell = torch.tensor([start], dtype=torch.int64, device="cuda")
    a = _lib.FlexWindowRefreshArgs()
    a.start, a.ring_rows = start_cell.data_ptr(), cap
    jobs = [(small, 0, act_w, 0, out["action"]), (small, act_w, n, 0, out["reward"]), (small, act_w + n, 1, 0, out["done"]),
            (nv, 0, n, row_off, out["nv"])]
    for j, (ring, col0, w, off, dst) in enumerate(jobs):
        a.base[j], a.dst[j], a.rows[j], a.row_off[j] = ring.data_ptr() + 4 * col0, dst.data_ptr(), rows, off
        a.width[j], a.src_stride[j] = w, ring.shape[1]
    a.n_jobs, a.n_cells, a.reward_job = len(jobs), 2, 1
    a.cell[0], a.cell_mod[0], a.cell[1], a.cell_mod[1] = cell[0:1].data_ptr(), cap, cell[1:2].data_ptr(), 1000
    ws = [torch.zeros(_lib.FLEXNET_TD_WS_FLOATS // 2, dtype=torch.float64, device="cuda") for _ in range(2)]
    t = _lib.FlexTdLossArgs()
    t.rows, t.n_agents, t.normalise, t.reward = rows, n, 1, out["reward"].data_ptr()
    t.workspace, t.workspace_floats = ws[0].data_ptr(), 2 * ws[0].numel()
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(lib.flexnet_window_refresh(C.byref(a), C.byref(t), stream), "flexnet_window_refresh")
    torch.cuda.synchronize()
    idx = (start + torch.arange(rows, device="cuda")) % cap
    assert torch.equal(out["action"], small[idx, :act_w]) and torch.equal(out["reward"], small[idx, act_w:act_w + n])
    assert torch.equal(out["done"], small[idx, act_w + n:act_w + n + 1]) and torch.equal(out["nv"], nv[(idx + row_off) % cap])
    assert cell.tolist() == [start % cap, start % 1000]
    t.workspace = ws[1].data_ptr()
    _lib.check(lib.flexnet_td_stats(C.byref(t), stream), "flexnet_td_stats")
    torch.cuda.synchronize()
    k = _lib.FLEXNET_TD_STAT_DOUBLES
    assert torch.equal(ws[0][:k], ws[1][:k])
    # without the rider; cells alone; refused arguments
    out["reward"].fill_(float("nan"))
    _lib.check(lib.flexnet_window_refresh(C.byref(a), None, stream), "flexnet_window_refresh")
    torch.cuda.synchronize()
    assert torch.equal(out["reward"], small[idx, act_w:act_w + n])
    a.n_jobs = 0
    cell.fill_(-1)
    _lib.check(lib.flexnet_window_refresh(C.byref(a), None, stream), "flexnet_window_refresh")
    torch.cuda.synchronize()
    assert cell.tolist() == [start % cap, start % 1000]
    assert lib.flexnet_window_refresh(C.byref(a), C.byref(t), stream) != 0           # the rider's job does not exist
    a.n_jobs, a.reward_job = len(jobs), 0
    assert lib.flexnet_window_refresh(C.byref(a), C.byref(t), stream) != 0           # job 0 is not the reward's copy
