"""Size-independent properties of the learner kernels at the update's full size (163 840 rows = 32 768 samples x 5
agents), where no CPU oracle finishes in seconds: permuting the samples permutes the per-row outputs bit for bit and
leaves the batch-reduced gradients unchanged up to summation order; weight gradients are additive over row splits and
exactly linear under power-of-two scaling."""
import types

import pytest
import torch

pytestmark = pytest.mark.gpu

B, N = 32768, 5


def _agent():
    from safe_marl_amd.nets import RNNAgent
    torch.manual_seed(0)
    args = types.SimpleNamespace(hid_size=64, layernorm=True, action_dim=4, agent_num=N, hid_activation="relu")
    agent = RNNAgent(144 + N, args).cuda()
    with torch.no_grad():
        for p in agent.parameters():
            p.copy_(torch.randn_like(p) * 0.3)
    return agent


def test_actor_kernel_is_equivariant_under_sample_permutation():
    from safe_marl_amd.nets import fused_actor_forward
    agent = _agent()
    g = torch.Generator(device="cuda").manual_seed(1)
    obs = torch.randn(B, N, 144, device="cuda", generator=g)
    hid = torch.randn(B, N, 64, device="cuda", generator=g)
    perm = torch.randperm(B, device="cuda", generator=g)
    with torch.no_grad():
        m1, h1 = fused_actor_forward(agent, obs, hid, N, True)
        m2, h2 = fused_actor_forward(agent, obs[perm].contiguous(), hid[perm].contiguous(), N, True)
    assert torch.equal(m1.view(B, N, 4)[perm], m2.view(B, N, 4))
    assert torch.equal(h1.view(B, N, 64)[perm], h2.view(B, N, 64))
    assert torch.isfinite(m1).all() and torch.isfinite(h1).all()


def test_critic_kernels_under_sample_permutation():
    from safe_marl_amd.nets import CriticTail, MLPCritic
    torch.manual_seed(2)
    args = types.SimpleNamespace(hid_size=64, layernorm=True, hid_activation="relu")
    c = MLPCritic(745, 1, args).cuda()
    with torch.no_grad():
        for p in c.parameters():
            p.copy_(torch.randn_like(p) * 0.2)
    g = torch.Generator(device="cuda").manual_seed(3)
    shared = torch.randn(B, 64, device="cuda", generator=g)
    ids = torch.randn(N, 64, device="cuda", generator=g)
    up = torch.randn(B, N, 1, device="cuda", generator=g)
    perm = torch.randperm(B, device="cuda", generator=g)
    params = [p for n, p in c.named_parameters() if not n.startswith("fc1")]
    out = []
    for s, u in ((shared, up), (shared[perm].contiguous(), up[perm].contiguous())):
        x = s.clone().requires_grad_(True)
        q = CriticTail.apply_composed(x, ids, c).view(B, N, 1)
        out.append((q.detach(), torch.autograd.grad((q * u).sum(), [x] + params)))
    assert torch.equal(out[0][0][perm], out[1][0])                           # per-row forward: bit for bit
    assert torch.equal(out[0][1][0][perm], out[1][1][0])                     # dz1 folded per sample: bit for bit
    for a, e in zip(out[0][1][1:], out[1][1][1:]):                           # sums over the batch: order changes only
        assert (a - e).abs().max().item() <= 2e-4 * max(1e-6, e.abs().max().item())


def test_weight_gradient_is_additive_and_linear():
    from safe_marl_amd.nets import tall_wgrad
    g = torch.Generator(device="cuda").manual_seed(4)
    k = B * N
    dy = torch.randn(k, 192, device="cuda", generator=g)
    x = torch.randn(k, 64, device="cuda", generator=g)
    full = tall_wgrad(dy, x)
    cut = 70003                                                               # an odd split, not a tile boundary
    parts = tall_wgrad(dy[:cut], x[:cut]) + tall_wgrad(dy[cut:], x[cut:])
    assert (full - parts).abs().max().item() <= 3e-7 * (k ** 0.5) * full.abs().max().item()
    assert torch.equal(tall_wgrad(dy * 4.0, x), full * 4.0)                   # power-of-two scaling is exact in fp32
    assert torch.equal(tall_wgrad(dy, x * 0.5), full * 0.5)
    cs = torch.empty(192, device="cuda")
    tall_wgrad(dy, x, colsum=cs)
    ones = tall_wgrad(dy, torch.ones(k, 1, device="cuda"))                    # the bias gradient is the product with a ones column
    assert (cs - ones[:, 0]).abs().max().item() <= 3e-7 * (k ** 0.5) * max(1.0, ones.abs().max().item())
