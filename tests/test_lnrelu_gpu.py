"""csrc/lnrelu.hip (include/flexnet.h: flexnet_lnrelu_forward / _backward) — relu(LayerNorm(z + bias + id column)) of
rnn_agent.py:25-29 with the one-hot id block of model.py:105-108 as a per-agent addend — against the PyTorch
composition and autograd; and the actor's update-batch forward built on it against the module path."""
import types

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _reference(z, bias, ids, ln_w, ln_b, n):
    x = z
    if bias is not None:
        x = x + bias
    if ids is not None:
        x = (x.view(-1, n, 64) + ids.unsqueeze(0)).reshape(-1, 64)
    if ln_w is not None:
        x = F.layer_norm(x, (64,), ln_w, ln_b, 1e-5)
    return torch.relu(x)


@pytest.mark.parametrize("rows,n,ln,with_bias,with_ids", [
    (163840, 5, True, True, True), (20480, 5, True, True, True), (4099 * 3, 3, True, True, True), (36, 4, False, True, True),
    (8, 8, True, False, True), (7, 1, True, True, False), (1, 1, True, True, True), (64, 2, False, False, False)])
def test_forward_and_backward_match_autograd(rows, n, ln, with_bias, with_ids):
    from safe_marl_amd.nets import _LnReluFn
    g = torch.Generator(device="cuda").manual_seed(rows + n)
    z = torch.randn(rows, 64, device="cuda", generator=g)
    up = torch.randn(rows, 64, device="cuda", generator=g)
    bias = torch.randn(64, device="cuda", generator=g) if with_bias else None
    ids = torch.randn(n, 64, device="cuda", generator=g) if with_ids else None
    ln_w = torch.randn(64, device="cuda", generator=g) if ln else None
    ln_b = torch.randn(64, device="cuda", generator=g) if ln else None
    res = []
    for fused in (True, False):
        leaves = [t.clone().requires_grad_(True) if t is not None else None for t in (z, bias, ids, ln_w, ln_b)]
        out = _LnReluFn.apply(*leaves, 1e-5, n) if fused else _reference(*leaves, n)
        grads = torch.autograd.grad((out * up).sum(), [t for t in leaves if t is not None])
        res.append((out,) + grads)
    for a, e in zip(*res):
        assert a.shape == e.shape
        tol = 2e-5 if a.shape[0] == rows and a.dim() == 2 and a.shape[1] == 64 and rows > 64 else 3e-4
        assert (a - e).abs().max().item() <= tol * max(1.0, e.abs().max().item()), a.shape


def test_backward_is_bit_reproducible():
    from safe_marl_amd.nets import _LnReluFn
    g = torch.Generator(device="cuda").manual_seed(0)
    z = torch.randn(50000, 64, device="cuda", generator=g)
    up = torch.randn(50000, 64, device="cuda", generator=g)
    p = [torch.randn(64, device="cuda", generator=g), torch.randn(5, 64, device="cuda", generator=g),
         torch.randn(64, device="cuda", generator=g), torch.randn(64, device="cuda", generator=g)]
    runs = []
    for _ in range(2):
        leaves = [t.clone().requires_grad_(True) for t in [z] + p]
        runs.append(torch.autograd.grad((_LnReluFn.apply(*leaves, 1e-5, 5) * up).sum(), leaves))
    for a, b in zip(*runs):
        assert torch.equal(a, b)


def test_policy_update_pass_matches_the_module_path():
    """Model.policy with gradients at an update batch (no id concat, fused epilogue, composed GRU cell, HIP weight
    gradients) against the module path the reference's layers define (fused_inference = False)."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
    from train_maddpg import DEFAULT_ALG_ARGS
    from safe_marl_amd.learner import MADDPG
    from safe_marl_amd.util import convert
    alg = dict(DEFAULT_ALG_ARGS)
    alg.update(alg="maddpg", agent_num=5, obs_size=144, state_size=110, action_dim=4)
    torch.manual_seed(3)
    model = MADDPG(convert(alg)).cuda()
    g = torch.Generator(device="cuda").manual_seed(4)
    b = 4096
    obs = torch.randn(b, 5, 144, device="cuda", generator=g)
    hid = torch.randn(b, 5, 64, device="cuda", generator=g)
    up_m = torch.randn(b, 5, 4, device="cuda", generator=g)
    up_h = torch.randn(b, 5, 64, device="cuda", generator=g)
    params = list(model.policy_dicts.parameters())

    def run():
        h0 = hid.clone().requires_grad_(True)
        means, _, hiddens = model.policy(obs, last_hid=h0)
        return (means, hiddens) + torch.autograd.grad((means * up_m).sum() + (hiddens * up_h).sum(), [h0] + params)

    fast = run()
    model.fused_inference = False
    plain = run()
    for a, e in zip(fast, plain):
        assert a.shape == e.shape
        assert (a - e).abs().max().item() <= 1e-4 * max(1.0, e.abs().max().item())
