"""csrc/optim.hip (include/flexnet.h: flexnet_clip_rmsprop) — clip_grad_norm_ + RMSprop.step() of
madrl/utils/trainer.py:34-35,86-90,103-107 in one launch — against the two PyTorch calls, over several steps, with the
clip active and inactive; the optimiser state stays PyTorch's."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _nets():
    torch.manual_seed(0)
    shapes = [(64, 149), (64,), (64,), (64,), (192, 64), (192, 64), (192,), (192,), (4, 64), (4,)]      # the actor's tensors
    a = [torch.nn.Parameter(torch.randn(s, device="cuda") * 0.1) for s in shapes]
    b = [torch.nn.Parameter(p.detach().clone()) for p in a]
    return a, b


@pytest.mark.parametrize("scale,max_norm", [(1.0, 1.0), (1e-3, 1.0), (5.0, 0.5)])
def test_matches_clip_grad_norm_and_rmsprop(scale, max_norm):
    from safe_marl_amd.optim import clip_and_step
    from torch.optim import RMSprop
    a, b = _nets()
    oa = RMSprop(a, lr=1e-3, alpha=0.99, eps=1e-5, capturable=True)
    ob = RMSprop(b, lr=1e-3, alpha=0.99, eps=1e-5, capturable=True)
    g = torch.Generator(device="cuda").manual_seed(1)
    for step in range(4):
        for pa, pb in zip(a, b):
            gr = torch.randn(pa.shape, device="cuda", generator=g) * scale
            pa.grad, pb.grad = gr.clone(), gr.clone()
        if step == 2:                      # a tensor without gradient is skipped by both
            a[3].grad = b[3].grad = None
        na = clip_and_step(oa, a, max_norm)
        nb = torch.nn.utils.clip_grad_norm_(b, max_norm)
        ob.step()
        assert abs(na.item() - nb.item()) <= 1e-6 * nb.item()
        for pa, pb in zip(a, b):
            assert (pa - pb).abs().max().item() <= 2e-6 * max(1.0, pb.abs().max().item())
            if pa.grad is not None:
                assert (pa.grad - pb.grad).abs().max().item() <= 2e-6 * max(1e-3, pb.grad.abs().max().item())
            sa, sb = oa.state[pa], ob.state[pb]
            assert set(sa) == set(sb)
            assert sa["step"].item() == sb["step"].item()
            assert (sa["square_avg"] - sb["square_avg"]).abs().max().item() <= 5e-6 * max(1e-12, sb["square_avg"].abs().max().item())
    # PyTorch continues from the state the kernel left
    sd = oa.state_dict()
    oc = RMSprop(a, lr=1e-3, alpha=0.99, eps=1e-5, capturable=True)
    oc.load_state_dict(sd)
    for pa in a:
        pa.grad = torch.ones_like(pa)
    oc.step()


def test_unsupported_configuration_falls_back():
    from safe_marl_amd.optim import clip_and_step
    from torch.optim import RMSprop
    a, b = _nets()
    oa = RMSprop(a, lr=1e-3, alpha=0.99, eps=1e-5, momentum=0.9, capturable=True)
    ob = RMSprop(b, lr=1e-3, alpha=0.99, eps=1e-5, momentum=0.9, capturable=True)
    for pa, pb in zip(a, b):
        pa.grad = torch.ones_like(pa)
        pb.grad = torch.ones_like(pb)
    clip_and_step(oa, a, 1.0)
    torch.nn.utils.clip_grad_norm_(b, 1.0)
    ob.step()
    for pa, pb in zip(a, b):
        assert torch.equal(pa, pb)
