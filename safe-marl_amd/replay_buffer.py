"""Device-resident replay memory with the reference's TransReplayBuffer surface
(utils/replay_buffer.py:3-30: ``add_experience``, ``get_batch``, ``clear``, ``.buffer`` whose only
use is ``len()``; trainer.py:31,67 and model.py:42,44,57 are the call sites).

Storage is a ring of pre-allocated tensors, one per Transition field (model.py:19), living where the
learner runs (HBM on the GPU box).  Semantics kept from the reference:
  * FIFO with capacity ``size``: the oldest transition is dropped first (replay_buffer.py:23-27);
  * a batch is ``batch_size`` CONSECUTIVE transitions starting at
    ``np.random.choice(len - batch_size + 1, 1, replace=False)[0]`` (replay_buffer.py:17-21),
    drawn from the global NumPy RNG so that ``np.random.seed`` reproduces the reference's index sequence.
Added for the vectorised path: ``add_batch`` appends one transition per environment in a single
indexed copy, and ``get_batch_tensors`` returns the window as device tensors (no host round trip,
no per-sample Python objects).  With N envs the ring is time-major — consecutive slots are the N envs
of one vector step — so a window is a contiguous, coalesced slab.
"""
from __future__ import annotations

from collections import namedtuple

import numpy as np
import torch as th

Transition = namedtuple("Transition", ("state", "action", "log_prob_a", "value", "next_value", "reward",
                                       "next_state", "done", "last_step", "action_avail", "last_hid", "hid"))

FIELDS = Transition._fields


class _LenView:
    """What ``.buffer`` exposes: the reference only ever calls len() on it."""

    def __init__(self, owner):
        self._owner = owner

    def __len__(self):
        return self._owner.length


class DeviceReplayBuffer:
    EXACT_STREAM_MAX = 16384       # the reference's buffer holds 5000 transitions (default.yaml:23)

    def __init__(self, size, device="cpu"):
        self.size = int(size)
        self.device = th.device(device)
        self.store = None          # dict field -> tensor [size, ...]
        self.head = 0              # physical slot of the oldest transition
        self.length = 0
        self.buffer = _LenView(self)
        self.consts = {}           # field -> python float, for fields that are never stored
        self.const_shapes = {}     # field -> trailing shape of the broadcast view

    # -- allocation on first use (shapes come from the first transition) ---------------------
    def _alloc(self, shapes):
        self.store = {k: th.zeros((self.size,) + tuple(s), dtype=th.float32, device=self.device)
                      for k, s in shapes.items()}

    def alloc_packed(self, shapes):
        """One [size, D] tensor holding every stored field side by side; ``store[field]`` are strided views into it.
        A whole transition batch then lands with ONE copy (``add_packed``) — what a graph-captured rollout needs —
        while ``add_batch`` / ``window`` keep working field by field."""
        widths = {k: int(np.prod(sh)) if len(sh) else 1 for k, sh in shapes.items()}
        self.packed_cols, off = {}, 0
        for k, w in widths.items():
            self.packed_cols[k] = (off, off + w, tuple(shapes[k]))
            off += w
        self.store2d = th.zeros(self.size, off, dtype=th.float32, device=self.device)
        self.store = {k: self.store2d[:, c0:c1].view((self.size,) + sh) for k, (c0, c1, sh) in self.packed_cols.items()}

    def record_views(self, rec):
        """Field views of a packed [B, D] staging record with the same column layout."""
        return {k: rec[:, c0:c1].view((rec.shape[0],) + sh) for k, (c0, c1, sh) in self.packed_cols.items()}

    def add_packed(self, rec):
        b = rec.shape[0]
        start = self._slots(b)
        first = min(b, self.size - start)
        self.store2d[start:start + first].copy_(rec[:first])
        if first < b:
            self.store2d[:b - first].copy_(rec[first:])
        return start

    def _slots(self, count):
        """Physical slots for ``count`` new transitions, dropping the oldest when full."""
        if count > self.size:
            raise ValueError("more transitions than capacity in one add")
        start = (self.head + self.length) % self.size
        overflow = max(0, self.length + count - self.size)
        self.head = (self.head + overflow) % self.size
        self.length = self.length + count - overflow
        return start

    def add_batch(self, **fields):
        """One transition per row: every field is a tensor [B, ...] already on the device.  Fields given as a
        python float are constants (e.g. action_avail = 1.0, the unused value/next_value = 0.0): nothing is stored
        for them, ``window`` hands out a broadcast view with the shape given in ``const_shapes``."""
        consts = {k: v for k, v in fields.items() if not isinstance(v, th.Tensor)}
        fields = {k: v for k, v in fields.items() if isinstance(v, th.Tensor)}
        b = next(iter(fields.values())).shape[0]
        if self.store is None:
            self._alloc({k: v.shape[1:] for k, v in fields.items()})
            self.consts = dict(consts)
        start = self._slots(b)
        first = min(b, self.size - start)
        for k, v in fields.items():
            dst = self.store[k]
            v = v.to(dst.dtype)
            dst[start:start + first].copy_(v[:first])
            if first < b:
                dst[:b - first].copy_(v[first:])

    def add_experience(self, trans):
        """replay_buffer.py:23-27 with the Transition of model.py:230-242 (numpy fields)."""
        row = {
            "state": np.asarray(trans.state, np.float32)[None],
            "action": np.asarray(trans.action, np.float32).reshape(1, *np.shape(trans.action)[-2:]),
            "log_prob_a": np.asarray(trans.log_prob_a, np.float32).reshape(1, *np.shape(trans.log_prob_a)[-2:]),
            "value": np.asarray(trans.value, np.float32).reshape(1, *np.shape(trans.value)[-2:]),
            "next_value": np.asarray(trans.next_value, np.float32).reshape(1, *np.shape(trans.next_value)[-2:]),
            "reward": np.asarray(trans.reward, np.float32)[None],
            "next_state": np.asarray(trans.next_state, np.float32)[None],
            "done": np.asarray([float(trans.done)], np.float32),
            "last_step": np.asarray([float(trans.last_step)], np.float32),
            "action_avail": np.asarray(trans.action_avail, np.float32).reshape(1, *np.shape(trans.action_avail)[-2:]),
            "last_hid": np.asarray(trans.last_hid, np.float32).reshape(1, *np.shape(trans.last_hid)[-2:]),
            "hid": np.asarray(trans.hid, np.float32).reshape(1, *np.shape(trans.hid)[-2:]),
        }
        self.add_batch(**{k: th.from_numpy(v).to(self.device) for k, v in row.items()})

    def clear(self):
        self.head = 0
        self.length = 0

    # -- sampling ------------------------------------------------------------------------------
    def sample_start(self, batch_size):
        sample_range = self.length - batch_size + 1
        if sample_range < 1:
            raise ValueError("not enough transitions for a batch")
        if sample_range <= self.EXACT_STREAM_MAX:
            # replay_buffer.py:18-19 verbatim, so that np.random.seed reproduces the reference's index sequence;
            # NumPy draws it as the head of a full permutation of `sample_range` elements
            return int(np.random.choice(sample_range, 1, replace=False)[0])
        # vectorised buffers hold 10^5..10^7 slots: the same uniform start without an O(n) permutation per sample
        return int(np.random.randint(sample_range))

    def window(self, start, batch_size):
        """Device tensors of logical transitions [start, start+batch_size)."""
        p0 = (self.head + start) % self.size
        out = {}
        for k, c in self.consts.items():
            shape = self.const_shapes.get(k, ())
            out[k] = th.full((1,) + tuple(1 for _ in shape), float(c), device=self.device).expand((batch_size,) + tuple(shape))
            out[k]._flex_const = float(c)          # readers may skip work that a known constant makes the identity
        if p0 + batch_size <= self.size:
            out.update({k: v[p0:p0 + batch_size] for k, v in self.store.items()})
        else:
            idx = (p0 + th.arange(batch_size, device=self.device)) % self.size
            out.update({k: v.index_select(0, idx) for k, v in self.store.items()})
        return Transition(**out)

    def get_batch_tensors(self, batch_size):
        return self.window(self.sample_start(batch_size), batch_size)

    def get_single(self, index):
        w = self.window(index, 1)
        return Transition(*[f[0] for f in w])

    def get_batch(self, batch_size):
        """replay_buffer.py:14-21: a list of ``batch_size`` per-transition tuples (device tensors inside).
        Kept for callers written against the reference (trainer.py:67-68 zips it back into columns)."""
        w = self.get_batch_tensors(batch_size)
        return [Transition(*[f[i] for f in w]) for i in range(batch_size)]


class TransReplayBuffer(DeviceReplayBuffer):
    """Name and constructor of utils/replay_buffer.py:3-6 (``TransReplayBuffer(int size)``)."""

    def __init__(self, size, device=None):
        if device is None:
            device = "cuda" if th.cuda.is_available() else "cpu"
        super().__init__(size, device)
