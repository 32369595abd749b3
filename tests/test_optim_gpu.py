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


@pytest.mark.parametrize("supported", [True, False])
def test_the_next_windows_refresh_rides_in_the_step(supported):
    """include/flexnet.h: flexnet_clip_rmsprop_refresh (optim.clip_and_step(refresh=...)) — the optimiser step of
    trainer.py:86-90 with the refresh of the NEXT sub-update's static batch (utils/replay_buffer.py:17-21, the statistics of
    model.py:308-323 included) riding in its two launches: parameters, optimiser state and norm equal the plain step's bit for
    bit, the batch / cell / statistics equal flexnet_window_refresh's.  A configuration the kernel does not cover takes the
    PyTorch step and launches the refresh on its own."""
    import ctypes as C
    from torch.optim import RMSprop
    from safe_marl_amd import _lib
    from safe_marl_amd.optim import clip_and_step
    lib = _lib.load()
    rows, n, act_w, stride, cap, start = 32768, 5, 20, 27, 200000, 200000 * 5 + 190000
    g = torch.Generator(device="cuda").manual_seed(7)
    small = torch.randn(cap, stride, device="cuda", generator=g)
    start_cell = torch.tensor([start], dtype=torch.int64, device="cuda")

    def refresh_args(ws):
        out = {k: torch.full((rows, w), float("nan"), device="cuda") for k, w in (("action", act_w), ("reward", n))}
        cell = torch.full((1,), -1, dtype=torch.int64, device="cuda")
        a = _lib.FlexWindowRefreshArgs()
        a.start, a.ring_rows = start_cell.data_ptr(), cap
        for j, (col0, w, dst) in enumerate(((0, act_w, out["action"]), (act_w, n, out["reward"]))):
            a.base[j], a.dst[j], a.rows[j], a.row_off[j] = small.data_ptr() + 4 * col0, dst.data_ptr(), rows, 0
            a.width[j], a.src_stride[j] = w, stride
        a.n_jobs, a.n_cells, a.reward_job = 2, 1, 1
        a.cell[0], a.cell_mod[0] = cell.data_ptr(), cap
        t = _lib.FlexTdLossArgs()
        t.rows, t.n_agents, t.normalise, t.reward = rows, n, 1, out["reward"].data_ptr()
        t.workspace, t.workspace_floats = ws.data_ptr(), 2 * ws.numel()
        return (a, t), out, cell

    kw = {} if supported else {"momentum": 0.9}
    res = []
    for ride in (True, False):
        a, _ = _nets()
        opt = RMSprop(a, lr=1e-3, alpha=0.99, eps=1e-5, capturable=True, **kw)
        gg = torch.Generator(device="cuda").manual_seed(3)
        for pa in a:
            pa.grad = torch.randn(pa.shape, device="cuda", generator=gg)
        ws = torch.zeros(_lib.FLEXNET_TD_WS_FLOATS // 2, dtype=torch.float64, device="cuda")
        refresh, out, cell = refresh_args(ws)
        if ride:
            norm = clip_and_step(opt, a, 1.0, refresh=refresh)
        else:
            norm = clip_and_step(opt, a, 1.0)
            _lib.check(lib.flexnet_window_refresh(C.byref(refresh[0]), C.byref(refresh[1]),
                                                  C.c_void_p(torch.cuda.current_stream().cuda_stream)), "flexnet_window_refresh")
        torch.cuda.synchronize()
        res.append(([p.detach().clone() for p in a], [opt.state[p]["square_avg"].clone() for p in a], norm.clone(),
                    out, cell.clone(), ws[:_lib.FLEXNET_TD_STAT_DOUBLES].clone()))
    (pa, va, na, oa, ca, wa), (pb, vb, nb, ob, cb, wb) = res
    assert torch.equal(na, nb) and all(torch.equal(x, y) for x, y in zip(pa, pb)) and all(torch.equal(x, y) for x, y in zip(va, vb))
    idx = (start + torch.arange(rows, device="cuda")) % cap
    for o in (oa, ob):
        assert torch.equal(o["action"], small[idx, :act_w]) and torch.equal(o["reward"], small[idx, act_w:act_w + n])
    assert ca.item() == cb.item() == start % cap
    assert torch.equal(wa, wb) and bool(wa.abs().sum() > 0)
    if supported:
        # the riding statistics blocks read the copy the riding copy blocks make: statistics arguments that name another
        # reward tensor are refused before any launch
        a, _ = _nets()
        opt = RMSprop(a, lr=1e-3, alpha=0.99, eps=1e-5, capturable=True)
        for pa in a:
            pa.grad = torch.ones_like(pa)
        (ra, rt), out, cell = refresh_args(torch.zeros(_lib.FLEXNET_TD_WS_FLOATS // 2, dtype=torch.float64, device="cuda"))
        other = torch.zeros(rows, n, device="cuda")
        rt.reward = other.data_ptr()
        with pytest.raises(RuntimeError):
            clip_and_step(opt, a, 1.0, refresh=(ra, rt))
        torch.cuda.synchronize()
