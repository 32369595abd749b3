"""GPU: the fused actor TRAINING pass (nets._ActorTrainFn: csrc/actor.hip forward with saved gates, csrc/gru.hip +
csrc/wgrad.hip + csrc/lnrelu.hip backward) against autograd on the PyTorch modules of rnn_agent.py:13-33 — forward
values, and the gradient of a random linear functional of the action means w.r.t. every actor parameter.  (Parity with
the reference itself: tests/test_learner_golden_gpu.py runs the same node inside the policy loss.)"""
import json
import os

import numpy as np
import pytest
import torch as th

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def _agent(**over):
    from safe_marl_amd.nets import RNNAgent
    from safe_marl_amd.util import convert
    d = json.load(open(os.path.join(G, "learner_args.json")))
    d.update(over)
    th.manual_seed(3)
    ag = RNNAgent(144 + 5, convert(d)).cuda()
    with th.no_grad():
        for p in ag.parameters():
            p.mul_(3.0).add_(0.05 * th.randn_like(p))       # the default init is tiny: make every term matter
    return ag


# fused_bwd: the first layer's backward inside the gate-gradient launch (round 3, gru_backward_fused_kernel) / the round-2
# composition (pointwise gate gradients + library GEMM + csrc/lnrelu.hip); 2565 rows: the last 16-row tile holds 5 rows
@pytest.mark.parametrize("fused_bwd", [True, False])
@pytest.mark.parametrize("rows,layernorm", [(2560, True), (10240, False), (163840, True), (20485, True), (2565, True)])
def test_fused_training_pass_matches_autograd(rows, layernorm, fused_bwd, monkeypatch):
    from safe_marl_amd import nets
    monkeypatch.setattr(nets, "GRU_BWD_FUSED", fused_bwd)
    n = 5
    ag = _agent(layernorm=layernorm)
    g = th.Generator(device="cuda").manual_seed(rows)
    obs = 0.5 * th.randn(rows, 144, device="cuda", generator=g)
    hid = 0.5 * th.randn(rows, 64, device="cuda", generator=g)
    proj = th.randn(rows, 4, device="cuda", generator=g) / rows

    def run(fused):
        ag.fused_training = fused
        ag.zero_grad()
        out = ag.forward_update(obs, hid, n, True)
        assert out is not None
        means, _, h = out
        (means * proj).sum().backward()
        return means.detach(), h.detach(), {k: p.grad.clone() for k, p in ag.named_parameters()}

    m1, h1, g1 = run(True) if rows % n == 0 else (None, None, None)
    if rows % n != 0:                                        # rows not a multiple of n_agents: the node declines, same numbers
        assert not nets.actor_train_supported(ag, obs, n, True)
        return
    m0, h0, g0 = run(False)
    assert (m1 - m0).abs().max().item() < 2e-5 * max(1.0, m0.abs().max().item())
    assert (h1 - h0).abs().max().item() < 2e-5
    for k in g0:
        ref = g0[k]
        tol = 2e-6 + 3e-4 * ref.abs().max().item()
        assert (g1[k] - ref).abs().max().item() < tol, (k, (g1[k] - ref).abs().max().item(), ref.abs().max().item())


def test_fused_training_pass_is_reproducible_and_used_by_policy():
    """Bit-reproducible (fixed-order reductions everywhere), and Model.policy's update pass really goes through the node."""
    from safe_marl_amd import nets
    from safe_marl_amd.learner import MADDPG
    from safe_marl_amd.util import convert
    d = json.load(open(os.path.join(G, "learner_args.json")))
    d.update(cuda=True)
    th.manual_seed(0)
    model = MADDPG(convert(d)).cuda()
    obs = th.randn(4096, 5, 144, device="cuda")
    hid = th.randn(4096, 5, 64, device="cuda")
    grads = []
    for _ in range(2):
        model.zero_grad()
        means, _, h = model.policy(obs, last_hid=hid)
        fn = means.grad_fn                                  # the [b, n, a] view of the node's output
        while fn is not None and "ActorTrainFn" not in type(fn).__name__ and fn.next_functions:
            fn = fn.next_functions[0][0]
        assert fn is not None and "ActorTrainFn" in type(fn).__name__
        means.square().sum().backward()
        grads.append([p.grad.clone() for p in model.policy_dicts.parameters()])
    for a, b in zip(*grads):
        assert th.equal(a, b)


@pytest.mark.parametrize("n,rows,act", [(3, 12288, 4), (1, 2048, 4), (8, 4096, 4), (5, 163840, 4), (5, 20480, 6), (2, 4098, 2)])
def test_first_layer_backward_in_the_gate_launch_equals_the_composition(n, rows, act, monkeypatch):
    """gru_backward_fused_kernel (dx on the matrix cores, LayerNorm / ReLU / bias / id columns backward, id-column sums written
    into fc1's gradient through strides) against the round-2 composition of the same node, for 1 / 2 / 3 / 5 / 8 agents, 2 / 4 / 6
    action columns and a batch whose last 16-row tile holds two rows: every
    parameter gradient to 2e-5 relative (fp32 sums in another order; see the note on ReLU's mask below), and twice the same bits."""
    from safe_marl_amd import nets
    from safe_marl_amd.nets import RNNAgent
    from safe_marl_amd.util import convert
    d = json.load(open(os.path.join(G, "learner_args.json")))
    d.update(agent_num=n, action_dim=act)
    th.manual_seed(5)
    ag = RNNAgent(144 + n, convert(d)).cuda()
    with th.no_grad():
        for p in ag.parameters():
            p.mul_(3.0).add_(0.05 * th.randn_like(p))
    g = th.Generator(device="cuda").manual_seed(rows + n)
    obs = 0.5 * th.randn(rows, 144, device="cuda", generator=g)
    hid = 0.5 * th.randn(rows, 64, device="cuda", generator=g)
    proj = th.randn(rows, act, device="cuda", generator=g) / rows

    def run(fused_bwd):
        monkeypatch.setattr(nets, "GRU_BWD_FUSED", fused_bwd)
        monkeypatch.setattr(nets, "_DEBUG_KEEP", {})
        ag.zero_grad()
        means, _, _ = ag.forward_update(obs, hid, n, True)
        (means * proj).sum().backward()
        return {k: p.grad.clone() for k, p in ag.named_parameters()}, nets._DEBUG_KEEP["dz"]

    (g1, dz1), (g1b, _), (g0, dz0) = run(True), run(True), run(False)
    # The fused kernel takes ReLU's mask from the forward's own output (save_x); the composition recomputes LayerNorm's output
    # and an output within an ulp of zero can land on the other side there (about 2e-7 per element: a row in some of these
    # cases — its whole dz changes, and with it that row's share of every first-layer gradient).  Rows with the same mask agree
    # to rounding; at most a couple of rows may differ, and only then do the parameter gradients get the wider bound.
    e = (dz1 - dz0).abs().amax(1)
    odd = int((e > 1e-7 + 2e-5 * dz0.abs().max().item()).sum().item())
    assert odd <= 2, odd
    for k in g0:
        assert th.equal(g1[k], g1b[k]), k
        ref = g0[k]
        err, scale = (g1[k] - ref).abs().max().item(), ref.abs().max().item()
        assert err < (1e-7 + 2e-5 * scale if odd == 0 else 0.1 * scale), (k, err, scale, odd)
