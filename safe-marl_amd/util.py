"""Action-selection helpers with the reference's names and semantics (utils/util.py), continuous
branch only — the flexibility-provision env is continuous (default.yaml:41).  Everything stays a
tensor on its device; nothing here forces a host sync except translate_action's numpy return, which
exists for the N=1 drop-in path (model.py:218 hands numpy to env.step)."""
from __future__ import annotations

from collections import namedtuple

import numpy as np
import torch as th
from torch.distributions.normal import Normal


def convert(dictionary):
    """util.py:190-191"""
    return namedtuple("GenericDict", dictionary.keys())(**dictionary)


def mean_all(x):
    """``x.mean()`` over all elements.  On the GPU, large inputs are summed in two steps (rows of 256, then the row
    sums): with this PyTorch-ROCm build a full ``mean`` / ``sum``-to-scalar of ~10^5+ elements is a multi-block kernel
    with a global semaphore, and such a scalar captured into a HIP graph came back stale or as a partial sum on replay
    (seen on the reported losses; the gradient of a mean is a broadcast and was never affected).  Same value up to
    fp32 summation order, same gradient."""
    n = x.numel()
    if x.is_cuda and n >= 16384 and n % 256 == 0:
        return x.reshape(-1, 256).sum(1).sum() / n
    return x.mean()


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
