"""Action-selection helpers with the reference's names and semantics (utils/util.py), continuous
branch only — the flexibility-provision env is continuous (default.yaml:41).  Everything stays a
tensor on its device; nothing here forces a host sync except translate_action's numpy return, which
exists for the N=1 drop-in path (model.py:218 hands numpy to env.step)."""
from __future__ import annotations

from collections import namedtuple

import numpy as np
import torch as th
from torch.distributions.normal import Normal


# HIP-graph captures are thread-local: with more than one rank ProcessGroupNCCL's watchdog thread polls events of
# outstanding collectives (hipEventQuery) from ITS thread, which the default "global" capture mode treats as an illegal
# call during capture and fails the capture with.  Only this thread's calls are checked.
CAPTURE_MODE = "thread_local"



class graph_capture:
    """``torch.cuda.graph(g, capture_error_mode=CAPTURE_MODE, **kw)`` with Python's cyclic garbage collector held off for
    the duration of the capture (and run once before it).  A dead reference cycle that owns an OLD ``CUDAGraph`` (a trainer
    that went out of scope: its nets, graphs and static batches point at each other) is freed whenever the collector
    happens to run; if that is in the middle of another capture, ``~CUDAGraph`` calls ``hipGraphDestroy`` on the capturing
    thread and the process dies with "operation not permitted when stream is capturing" (seen in bench.py's sixth training
    leg: a trainer per leg).  This PyTorch build no longer collects at capture begin by itself."""

    def __init__(self, graph, collect=True, **kw):
        self.ctx = th.cuda.graph(graph, capture_error_mode=CAPTURE_MODE, **kw)
        self.collect = collect                        # False: the caller collected already (a run of captures in a row)

    def __enter__(self):
        import gc
        if self.collect:
            gc.collect()
        self.was_enabled = gc.isenabled()
        gc.disable()
        try:
            return self.ctx.__enter__()
        except BaseException:
            if self.was_enabled:
                gc.enable()
            raise

    def __exit__(self, *exc):
        import gc
        try:
            return self.ctx.__exit__(*exc)
        finally:
            if self.was_enabled:
                gc.enable()


FALLBACKS = {}          # reason -> count: fused HIP paths that declined on GPU tensors (visible, never silent)


def note_fallback(key, detail=""):
    """A fused HIP path declined a GPU call (shape or configuration outside what the kernel covers) and the PyTorch
    composition runs instead: 4-5x slower, same numbers.  Counted in ``FALLBACKS`` and reported ONCE per reason, so a
    configuration drift (say hid_size != 64) cannot quietly lose the fast path under green tests."""
    first = key not in FALLBACKS
    FALLBACKS[key] = FALLBACKS.get(key, 0) + 1
    if first:
        import warnings
        warnings.warn(f"safe_marl_amd: fused path '{key}' declined ({detail}); the PyTorch composition runs instead",
                      RuntimeWarning, stacklevel=3)


# kernels that must never be recorded into a HIP graph with this PyTorch-ROCm build: ATen's multi-block reductions
# keep a global semaphore that a captured memset node resets, and replays came back stale or partial
GRAPH_DENYLIST = ("at::native::reduce_kernel", "batch_norm_collect_statistics", "batch_norm_backward_reduce",
                  "at::native::(anonymous namespace)::LpNormFunctor")


def audit_graph_body(fn, allow=()):
    """Run ``fn`` (the body about to be captured) eagerly under torch.profiler and return the device kernel names it
    launched; raises if one of them is on ``GRAPH_DENYLIST``.  Used by the tests and, with FLEX_GRAPH_AUDIT=1, by the
    capture sites themselves (trainer._capture_sub_update, learner.RolloutGraph.capture)."""
    from torch.profiler import ProfilerActivity, profile
    th.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        fn()
        th.cuda.synchronize()
    names = sorted({ev.key for ev in prof.key_averages() if "cuda" in str(getattr(ev, "device_type", "")).lower()})
    bad = [n for n in names if any(d in n for d in GRAPH_DENYLIST) and not any(a in n for a in allow)]
    if bad:
        raise RuntimeError("graph body launches multi-block ATen reductions (stale on HIP-graph replay): " + "; ".join(bad))
    return names


def convert(dictionary):
    """util.py:190-191"""
    return namedtuple("GenericDict", dictionary.keys())(**dictionary)


_SUM_WS = {}
_UNIT = {}
_CONST_GRADS = {}


def unit_seed(device):
    """The scalar 1.0 a backward pass starts from, one per device and never written again.  Handed to autograd as the root
    gradient (``grad_outputs``) it saves the engine's own ones-fill; the loss nodes below recognise it BY ADDRESS
    (``is_unit_seed``) and return their stored / constant gradient as it is instead of multiplying it by one — two to
    three ~5 us launches per sub-update that carried no information."""
    device = th.device(device)
    if device.type == "cuda" and device.index is None:
        device = th.device("cuda", th.cuda.current_device())
    if device not in _UNIT:
        _UNIT[device] = th.ones((), dtype=th.float32, device=device)
    return _UNIT[device]


def is_unit_seed(g):
    u = _UNIT.get(g.device)
    return u is not None and g.dim() == 0 and g.data_ptr() == u.data_ptr()


def const_grad(shape, value, device):
    """A read-only tensor of ``shape`` filled with ``value``, cached (the gradient of a mean under the unit seed).  Made
    while a HIP graph is being captured it is not cached: its memory and its fill would belong to that graph.  Entries
    are NEVER evicted: a sub-update graph captured after its eager warm-up has the cached tensor's address baked in, and
    the cache is that tensor's only owner — one small tensor per (shape, value) a process ever asks for."""
    key = (th.device(device), tuple(shape), float(value))
    t = _CONST_GRADS.get(key)
    if t is None:
        t = th.full(tuple(shape), float(value), dtype=th.float32, device=device)
        if not (t.is_cuda and th.cuda.is_current_stream_capturing()):
            _CONST_GRADS[key] = t
    return t


class _MeanAllFn(th.autograd.Function):
    """x.mean() over all elements through csrc/tdloss.hip's fixed-order sum (include/flexnet.h: flexnet_scaled_sum);
    the gradient of a mean is a broadcast."""

    @staticmethod
    def forward(ctx, x, sign):
        import ctypes as C
        from . import _lib
        lib = _lib.load()
        flat = x.contiguous().view(-1)
        out = th.empty((), dtype=th.float32, device=x.device)
        if x.device not in _SUM_WS:
            _SUM_WS[x.device] = th.empty(_lib.FLEXNET_SUM_WS_FLOATS // 2, dtype=th.float64, device=x.device)
        ws = _SUM_WS[x.device]
        a = _lib.FlexSumArgs()
        a.n, a.scale = flat.numel(), sign / flat.numel()
        a.x, a.out, a.workspace, a.workspace_floats = flat.data_ptr(), out.data_ptr(), ws.data_ptr(), 2 * ws.numel()
        _lib.check(lib.flexnet_scaled_sum(C.byref(a), C.c_void_p(th.cuda.current_stream().cuda_stream)), "flexnet_scaled_sum")
        ctx.shape, ctx.scale = x.shape, sign / flat.numel()
        return out

    @staticmethod
    def backward(ctx, g):
        if is_unit_seed(g):
            return const_grad(ctx.shape, ctx.scale, g.device), None
        return (g * ctx.scale).expand(ctx.shape), None


def mean_all(x, sign=1.0):
    """``x.mean()`` over all elements.  On the GPU it is this project's fixed-order sum kernel: with this PyTorch-ROCm
    build a full ``mean`` / ``sum``-to-scalar of ~10^5+ elements is a multi-block kernel with a global semaphore, and such
    a scalar captured into a HIP graph came back stale or as a partial sum on replay (seen on the reported losses; the
    gradient of a mean is a broadcast and was never affected).  Same value up to fp32 summation order (the kernel
    accumulates in fp64), same gradient."""
    if x.is_cuda and x.dtype == th.float32 and x.numel() >= 1:
        return _MeanAllFn.apply(x, float(sign))      # ``sign`` = -1: mean(-x) without a negation pass over x
    return x.mean() if sign == 1.0 else (sign * x).mean()


def normal_entropy(mean, std):
    """util.py:35-36"""
    return mean_all(Normal(mean, std, validate_args=False).entropy())


def select_action(args, logits, status="train", exploration=True, info={}):
    """util.py:50-85 (continuous branch).

    train+explore with action_enforcebound: y = tanh(x), x ~ N(mean, std) reparameterised, and
    log_prob = log N(x) - log(1 - y^2 + 1e-6) (util.py:57-64); without the bound: mean + (clipped) noise
    (util.py:66-74); train without exploration: the raw mean (util.py:75-77); test: tanh(mean) when the
    bound is enforced (util.py:79-82)."""
    if not args.continuous:
        raise NotImplementedError("discrete control is outside the flexibility-provision hot path")
    act_mean = logits
    if status == "train":
        if not exploration:
            return act_mean, None
        act_std = info["log_std"].exp()
        if args.action_enforcebound:
            normal = Normal(act_mean, act_std, validate_args=False)   # validation is a device reduction + host sync
            x_t = normal.rsample()
            y_t = th.tanh(x_t)
            log_prob = normal.log_prob(x_t) - th.log(1 - y_t.pow(2) + 1e-6)
            return y_t, log_prob
        normal = Normal(th.zeros_like(act_mean), act_std, validate_args=False)
        x_t = normal.rsample()
        log_prob = normal.log_prob(x_t)
        if info.get("clip", False):
            return act_mean + th.clamp(x_t, min=-args.clip_c, max=args.clip_c), log_prob
        return act_mean + x_t, log_prob
    if status == "test":
        return (th.tanh(act_mean) if args.action_enforcebound else act_mean), None
    raise ValueError(status)


def scale_action(args, action):
    """The arithmetic of translate_action (util.py:125-128) on device: clamp to [low, high], then
    0.5*(a+1)*(high-low)+low — with low=0, high=1 the env sees [0.5, 1.0]."""
    low, high = args.action_low, args.action_high
    return 0.5 * (th.clamp(action, min=low, max=high) + 1.0) * (high - low) + low


def translate_action(args, action, env):
    """util.py:121-130: (squeezed policy action, numpy env action)."""
    if not args.continuous:
        raise NotImplementedError
    actions = action.detach().squeeze()
    return actions, scale_action(args, actions).cpu().numpy()


def prep_obs(state=[]):
    """util.py:135-145: list-of-arrays observation(s) -> float32 tensor with a leading batch axis."""
    state = np.array(state)
    if state.ndim == 2:
        state = np.stack(state, axis=0)
    elif state.ndim == 4:
        state = np.concatenate(state, axis=0)
    else:
        raise RuntimeError("The shape of the observation is incorrect.")
    return th.tensor(state).float()


def get_grad_norm(args, params):
    """util.py:159-161: clip_grad_norm_ — clips in place and returns the pre-clip total norm."""
    return th.nn.utils.clip_grad_norm_(params, args.grad_clip_eps)


def merge_dict(stat, key, value):
    stat[key] = stat.get(key, 0) + value


def dict2str(d, name):
    return "\n".join([f"{name}:"] + [f"\t{k}: {v}" for k, v in d.items()])
