"""clip_grad_norm_ + torch.optim.RMSprop.step() as two small HIP launches (csrc/optim.hip, include/flexnet.h:
flexnet_clip_rmsprop) for the two small networks of the MADDPG path (madrl/utils/trainer.py:34-35,86-90,103-107).

The optimiser object stays a ``torch.optim.RMSprop``: its ``state`` (``square_avg``, ``step``) is created exactly as
PyTorch creates it and updated in place, so ``state_dict()`` round-trips and a later ``optimizer.step()`` continues
from it.  Configurations the kernel does not cover fall back to the two PyTorch calls."""
from __future__ import annotations

import ctypes as C

import torch as th


def _supported(opt, params):
    from . import _lib
    if len(opt.param_groups) != 1:
        return False
    g = opt.param_groups[0]
    if g.get("momentum", 0) != 0 or g.get("centered", False) or g.get("weight_decay", 0) != 0 or g.get("maximize", False):
        return False
    if not g.get("capturable", False) or th.is_tensor(g["lr"]):
        return False
    live = [p for p in params if p.grad is not None]
    if len(live) > _lib.FLEXNET_OPT_MAX_TENSORS or sum(p.numel() for p in live) > _lib.FLEXNET_OPT_MAX_ELEMENTS:
        return False
    return all(p.is_cuda and p.dtype == th.float32 and p.is_contiguous() and p.grad.dtype == th.float32
               and p.grad.is_contiguous() and not p.grad.is_sparse for p in live)


def clip_and_step(opt, params, max_norm, refresh=None):
    """Clip the gradients of ``params`` to ``max_norm`` (2-norm over all of them), take one RMSprop step, return the
    pre-clip norm (a 0-dim tensor) — ``clip_grad_norm_`` + ``opt.step()``.
    ``refresh`` = (FlexWindowRefreshArgs, FlexTdLossArgs or None): the refresh of the NEXT sub-update's static batch rides in
    the step's two launches (flexnet_clip_rmsprop_refresh; trainer._ensure_event_graph) — launched on its own after the step
    where the kernel path does not apply."""
    if not (params and params[0].is_cuda and _supported(opt, params)):
        if params and params[0].is_cuda:
            from .util import note_fallback
            note_fallback("clip_rmsprop", "optimiser configuration outside csrc/optim.hip (capturable RMSprop, no momentum / "
                                          "centering / weight decay, contiguous fp32 tensors)")
        norm = th.nn.utils.clip_grad_norm_(params, max_norm)
        opt.step()
        if refresh is not None:
            _launch_refresh(refresh)
        return norm
    from . import _lib
    lib = _lib.load()
    g = opt.param_groups[0]
    a = _lib.FlexClipRmspropArgs()
    a.lr, a.alpha, a.eps, a.max_norm = float(g["lr"]), float(g["alpha"]), float(g["eps"]), float(max_norm)
    scratch = th.empty(1 + 64, dtype=th.float32, device=params[0].device)      # [norm | FLEXNET_OPT_WS_FLOATS partials]
    norm = scratch[0]
    a.total_norm, a.workspace = norm.data_ptr(), scratch[1:].data_ptr()
    k = 0
    for p in params:
        if p.grad is None:
            continue
        st = opt.state[p]
        if len(st) == 0:                       # torch/optim/rmsprop.py _init_group, capturable
            st["step"] = th.zeros((), dtype=th.float32, device=p.device)
            st["square_avg"] = th.zeros_like(p, memory_format=th.preserve_format)
        a.numel[k], a.param[k], a.grad[k] = p.numel(), p.data_ptr(), p.grad.data_ptr()
        a.square_avg[k], a.step[k] = st["square_avg"].data_ptr(), st["step"].data_ptr()
        k += 1
    a.n_tensors = k
    stream = C.c_void_p(th.cuda.current_stream().cuda_stream)
    if refresh is not None and k > 0:
        ra, td = refresh
        _lib.check(lib.flexnet_clip_rmsprop_refresh(C.byref(a), C.byref(ra), C.byref(td) if td is not None else None, stream),
                   "flexnet_clip_rmsprop_refresh")
        return norm
    _lib.check(lib.flexnet_clip_rmsprop(C.byref(a), stream), "flexnet_clip_rmsprop")
    if refresh is not None:
        _launch_refresh(refresh)
    return norm


def _launch_refresh(refresh):
    from . import _lib
    ra, td = refresh
    _lib.check(_lib.load().flexnet_window_refresh(C.byref(ra), C.byref(td) if td is not None else None,
                                                  C.c_void_p(th.cuda.current_stream().cuda_stream)), "flexnet_window_refresh")
