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

    # -- slab mode: the vectorised, graph-captured rollout ------------------------------------------------------------
    ROW_W = 8                      # floats per row record: [Pd, Qd, Ppv, V, price, E, older, 0] (include/flexenv.h)
    HID_TAIL_SLABS = 40            # mirrored tail of the hidden-state ring: windows of up to 39 slabs (+ next) are read in place

    def alloc_slabs(self, n_envs, n_agents, obs_dim, act_dim, hid_dim, history=None):
        """Slab-structured ring for N environments stepping in lockstep (include/flexnet.h: flexnet_rollout_pack writes it
        at a device-side cursor).  Slab k = vector step k: the observation acted on, ``hid_ring[k]`` the hidden state going
        in, ``small_ring[k]`` = [action | reward | done | last_step].  Every observation is stored once: next_state of slab
        k is the observation of slab k + 1, model.py:241's ``hid`` is ``hid_ring[k + 1]``.

        ``history`` = H (row mode; the flexibility-provision env's observation is its last H feature rows, env:387-401): the
        ring keeps every feature ROW once — ``row_ring[k]`` = one 32-byte record per (env, agent), written by the env's
        step kernel (FLEX_STEP_OBS_RING) — 1/18 of the bytes of a stacked observation per slab; a window's stacked
        observations are formed when it is gathered (flexnet_gather_window: the gather is the im2col).  The H - 1 slabs
        behind the oldest transition stay in the ring for that (H - 1 physical slabs on top of ``size // n_envs``).
        ``history`` None: ``obs_ring[k]`` holds the stacked observation itself (general rollout bodies, hand-filled rings).

        Host-side bookkeeping mirrors the device cursor: slab counter ``k`` (monotone), the slab being filled is
        ``k % slabs``; transitions are the COMPLETE slabs still in the ring.  A slab left without an action (the rollout
        was restarted from a hard reset instead of continuing) is a *gap*; sampled windows never span one."""
        self.n_envs, self.n_agents, self.obs_dim, self.act_dim, self.hid_dim = n_envs, n_agents, obs_dim, act_dim, hid_dim
        self.history = int(history) if history else None
        if self.history is not None and (obs_dim % self.history != 0 or obs_dim // self.history != 6):
            raise ValueError("row mode stores 6-feature rows: obs_dim must be 6 * history")
        self.keep_back = self.history - 1 if self.history else 0      # slabs behind a transition its observation reaches into
        self.slabs = self.size // n_envs + self.keep_back
        if self.slabs - self.keep_back < 4:
            raise ValueError("slab replay needs room for at least 4 vector steps")
        no, nh = n_agents * obs_dim, n_agents * hid_dim
        self.small_w = ((n_agents * act_dim + n_agents + 2 + 3) // 4) * 4       # 16-byte rows
        dev = self.device
        if self.history is None:
            self.obs_ring = th.zeros(self.slabs, n_envs, no, dtype=th.float32, device=dev)
            self.row_ring = None
        else:
            self.obs_ring = None
            self.row_ring = th.zeros(self.slabs, n_envs, n_agents * self.ROW_W, dtype=th.float32, device=dev)
        # (the hidden-state ring has room for a mirrored tail behind its last slab — enable_stacked_ring uses it; everybody else
        #  sees the ring proper)
        self.hid_tail_slabs = min(self.slabs, self.HID_TAIL_SLABS) if self.history is not None else 0
        self.hid_store = th.zeros(self.slabs + self.hid_tail_slabs, n_envs, nh, dtype=th.float32, device=dev)
        self.hid_ring = self.hid_store[:self.slabs]
        self.small_ring = th.zeros(self.slabs, n_envs, self.small_w, dtype=th.float32, device=dev)
        # bootstrap values r + gamma (1 - done) Q'(s', pi(s')) needs, per transition, filed by the trainer for the windows of
        # ONE update event (trainer.replay_event: the networks behind them do not change between its value sub-updates)
        self.nv_ring = th.zeros(self.slabs, n_envs, n_agents, dtype=th.float32, device=dev)
        # physical slab indices on the device (include/flexnet.h): [0] the slab the policy reads, [1] the slab being filled
        self.cursor = th.zeros(2, dtype=th.int64, device=dev)
        self.k = 0                   # host mirror of cursor[0]
        self.first = 0               # oldest slab counter whose transition may still be in the ring
        self.gaps = []               # slab counters without a transition (half-written by the step before a hard reset)
        self.consts = {"log_prob_a": 0.0, "value": 0.0, "next_value": 0.0, "action_avail": 1.0}
        self.const_shapes = {"log_prob_a": (n_agents, act_dim), "value": (n_agents, 1), "next_value": (n_agents, 1),
                             "action_avail": (n_agents, act_dim)}
        self.store = None
        self.head = 0
        self.length = 0

    @property
    def slab_mode(self):
        return getattr(self, "hid_ring", None) is not None

    @property
    def row_mode(self):
        return getattr(self, "row_ring", None) is not None

    def release_slabs(self):
        """Back to the field-by-field mode (the graph rollout could not be captured): the rings and their bookkeeping go,
        ``add_batch`` allocates its own store on the next call.  Transitions the ring held are dropped."""
        self.obs_ring = self.row_ring = self.hid_ring = self.hid_store = self.small_ring = self.nv_ring = self.cursor = None
        self.stack_ring = None
        self.k = self.first = 0
        self.gaps = []
        self.consts, self.const_shapes = {}, {}
        self.store = None
        self.head = self.length = 0

    def begin_stream(self, first_obs):
        """A rollout starts from a hard reset: ``first_obs`` [N, n, obs] becomes the observation of the slab at the
        cursor, its hidden state is zero.  If a previous stream left the slab at the cursor half-written (observation
        without an action), that slab becomes a gap and the stream starts one slab later."""
        if self.k > 0:
            self.gaps.append(self.k)
            self.k += 1
        p = self.k % self.slabs
        if self.row_mode:
            # the first observation of an episode: its newest feature row, no older ones (zero left-padding)
            rec = self.row_ring[p].view(self.n_envs, self.n_agents, self.ROW_W)
            rec.zero_()
            rec[:, :, :6].copy_(first_obs.reshape(self.n_envs, self.n_agents, self.obs_dim)[:, :, -6:])
        else:
            self.obs_ring[p].copy_(first_obs.reshape(self.n_envs, -1))
        self.hid_ring[p].zero_()
        self.cursor.fill_(p)
        self._retire()

    def stepped(self):
        """Host mirror of one flexnet_rollout_pack launch (or graph replay): slab k is complete, the cursor moved on.
        Returns the physical index of the slab just completed."""
        done_slab = self.k % self.slabs
        self.k += 1
        self._retire()
        return done_slab

    def _retire(self):
        # the slab at the cursor (and its half-written successor after the next step) overwrite the oldest ones; in row mode
        # the oldest transition's observation also reaches keep_back slabs further back
        self.first = max(self.first, self.k + 2 - self.slabs + self.keep_back)
        self.gaps = [g for g in self.gaps if g >= self.first]
        self.length = self.n_envs * (self.k - self.first - len(self.gaps))

    def _runs(self):
        """Maximal runs [a, b) of slab counters holding consecutive complete transitions."""
        runs, a = [], self.first
        for g in self.gaps:
            if g > a:
                runs.append((a, g))
            a = g + 1
        if self.k > a:
            runs.append((a, self.k))
        return runs

    def sample_slot(self, batch_size):
        """utils/replay_buffer.py:17-21 on the slab ring: a uniformly random start among the windows of ``batch_size``
        consecutive transitions (slots; time-major, so consecutive slots are the N environments of one vector step, then
        the next step).  Returns the start as a global slot number (slab counter * N + env)."""
        N = self.n_envs
        spans = [(a * N, (b - a) * N - batch_size + 1) for a, b in self._runs() if (b - a) * N >= batch_size]
        total = sum(c for _, c in spans)
        if total < 1:
            raise ValueError("not enough transitions for a batch")
        r = int(np.random.randint(total))
        for base, c in spans:
            if r < c:
                return base + r
            r -= c
        raise AssertionError

    def warmup_slot(self, rows):
        """Start of a window of ``rows`` consecutive complete slots for a capture's warm-up steps: the first run of the
        ring that is long enough (the oldest run otherwise — its tail may then be slabs no transition was filed in, which
        the warm-up, whose steps are undone, only needs to be finite).  Deterministic: sample_slot would move the global
        NumPy stream that utils/replay_buffer.py:17-21 draws the real windows from."""
        N = self.n_envs
        for a, b in self._runs():
            if (b - a) * N >= rows:
                return a * N
        return self._logical_to_slot(0)

    def segments(self, slot, rows):
        """Physical pieces of the global slot range [slot, slot + rows): [(physical_slot, count)] (two at the ring's seam)."""
        cap = self.slabs * self.n_envs
        p = slot % cap
        first = min(rows, cap - p)
        return [(p, first)] + ([(0, rows - first)] if first < rows else [])

    def slab_window(self, slot, batch_size):
        """Transition of device tensors for the window starting at global slot ``slot`` (eager consumers)."""
        N, n = self.n_envs, self.n_agents

        def take(ring, start, rows):
            flat = ring.view(self.slabs * N, -1)
            parts = [flat[p:p + c] for p, c in self.segments(start, rows)]
            return parts[0] if len(parts) == 1 else th.cat(parts)

        na = n * self.act_dim
        small = take(self.small_ring, slot, batch_size)
        if self.row_mode:
            state, next_state = self.stacked_obs(slot, batch_size), self.stacked_obs(slot + N, batch_size)
        else:
            state, next_state = take(self.obs_ring, slot, batch_size), take(self.obs_ring, slot + N, batch_size)
        out = {"state": state.view(batch_size, n, self.obs_dim),
               "next_state": next_state.view(batch_size, n, self.obs_dim),
               "last_hid": take(self.hid_ring, slot, batch_size).view(batch_size, n, self.hid_dim),
               "hid": take(self.hid_ring, slot + N, batch_size).view(batch_size, n, self.hid_dim),
               "action": small[:, :na].reshape(batch_size, n, self.act_dim),
               "reward": small[:, na:na + n], "done": small[:, na + n], "last_step": small[:, na + n + 1]}
        for k, c in self.consts.items():
            shape = self.const_shapes.get(k, ())
            out[k] = th.full((1,) + tuple(1 for _ in shape), float(c), device=self.device).expand((batch_size,) + tuple(shape))
            out[k]._flex_const = float(c)
        return Transition(**out)

    def stacked_obs(self, slot, rows, out=None):
        """Row mode: the stacked observations [rows, n_agents * obs_dim] of the global slot range [slot, slot + rows), formed
        from the row ring by ONE launch of flexnet_gather_window (include/flexnet.h); the ring's seam is the kernel's business."""
        import ctypes as C
        from . import _lib
        if out is None:
            out = th.empty(rows, self.n_agents * self.obs_dim, dtype=th.float32, device=self.device)
        a = _lib.FlexWindowArgs()
        a.row_ring, a.dst, a.rows, a.first_slot = self.row_ring.data_ptr(), out.data_ptr(), rows, int(slot)
        a.n_envs, a.n_agents, a.history, a.slabs = self.n_envs, self.n_agents, self.history, self.slabs
        _lib.check(_lib.load().flexnet_gather_window(C.byref(a), C.c_void_p(th.cuda.current_stream().cuda_stream)),
                   "flexnet_gather_window")
        return out

    # -- stacked-observation ring (row mode, round 5): every slab's stacked observations formed ONCE ------------------------
    def enable_stacked_ring(self, max_window_rows):
        """Keep, next to the row ring, the stacked observations [n_agents * obs_dim] of every slab — ``stack_ring`` [slabs * N
        + tail, n * obs] — so that a sampled window is a contiguous run of its rows and captured sub-updates read it IN PLACE
        (nets.RING_VIEWS) instead of gathering it into a static batch first.  A slab is expanded once, when it has entered the
        replay (``expand_stacked``: flexnet_gather_window into the ring); at the reference's sample reuse the per-window gather
        expanded each observation ~6 times.  The first ``tail`` rows are mirrored behind the ring's end, so a window of up to
        ``max_window_rows`` rows that starts anywhere in the ring never wraps.  2 880 B per transition: 2.5 GB for the bench's
        ring of 192 + 23 slabs of 4 096 environments."""
        if not self.row_mode:
            raise RuntimeError("the stacked-observation ring lives next to the ROW ring (row mode)")
        N = self.n_envs
        tail = -(-int(max_window_rows) // N) * N
        if getattr(self, "stack_ring", None) is not None and self.stack_tail >= tail:
            return
        if tail > self.slabs * N:
            raise ValueError("window longer than the ring")
        # (a longer tail than the existing ring's: a NEW ring — graphs captured against the old one have its address baked in
        #  and are recaptured when next used, trainer._ensure_graph compares ``stack_gen``)
        self.stack_gen = getattr(self, "stack_gen", 0) + 1
        old = getattr(self, "stack_ring", None)
        if old is not None:
            from . import nets
            old_ptr = old.untyped_storage().data_ptr()
            for key in list(nets.RING_VIEWS):
                ring, _cell, ref = nets.RING_VIEWS[key]
                if ref() is None or ring.untyped_storage().data_ptr() == old_ptr:
                    del nets.RING_VIEWS[key]
            del old
        self.stack_rows = self.slabs * N
        self.stack_tail = tail
        self.stack_ring = th.zeros(self.stack_rows + tail, self.n_agents * self.obs_dim, dtype=th.float32, device=self.device)
        self.stacked_next = None         # slab counter the next expansion starts at (None: the oldest transition's)
        # every possible first row, as int64: a window's cell is set by ONE 8-byte row copy inside the gather launch that
        # refreshes the batch's other fields (no launch of its own: a fill kernel per cell was 4.5 us of a 360-us sub-update)
        self.slot_table = th.arange(self.stack_rows, dtype=th.int64, device=self.device)

    def expand_stacked(self):
        """Bring the stacked ring up to date: the slabs filed since the last call (the slab at the cursor included — it holds
        the newest observation, the next_state of the last complete transition)."""
        if getattr(self, "stack_ring", None) is None:
            return
        N = self.n_envs
        lo = self.first if self.stacked_next is None else max(self.stacked_next, self.first)
        hi = self.k
        if hi - lo + 1 > self.slabs:
            lo = hi - self.slabs + 1
        tail_slabs = self.stack_tail // N
        c = lo
        while c <= hi:
            p = c % self.slabs
            run = min(hi - c + 1, self.slabs - p)
            self.stacked_obs(c * N, run * N, out=self.stack_ring[p * N:(p + run) * N])
            if p < tail_slabs:                            # the ring's head, mirrored behind its end
                q = min(run, tail_slabs - p)
                self.stack_ring[self.stack_rows + p * N:self.stack_rows + (p + q) * N].copy_(self.stack_ring[p * N:(p + q) * N])
            if p < self.hid_tail_slabs:                   # ... and the hidden states of the same slabs (read in place as well)
                q = min(run, self.hid_tail_slabs - p)
                self.hid_store[self.slabs + p:self.slabs + p + q].copy_(self.hid_store[p:p + q])
            c += run
        self.stacked_next = hi + 1

    @property
    def obs_source_ring(self):
        """Name of the ring observations are gathered from: "row_ring" (row mode) or "obs_ring"."""
        return "row_ring" if self.row_mode else "obs_ring"

    # what a field of the Transition is in the ring: (ring name, first column, width or None = whole row, slab offset)
    def field_source(self, name):
        n, na = self.n_agents, self.n_agents * self.act_dim
        obs = self.obs_source_ring
        return {"state": (obs, 0, None, 0), "next_state": (obs, 0, None, 1),
                "last_hid": ("hid_ring", 0, None, 0), "hid": ("hid_ring", 0, None, 1),
                "action": ("small_ring", 0, na, 0), "reward": ("small_ring", na, n, 0),
                "done": ("small_ring", na + n, 1, 0), "last_step": ("small_ring", na + n + 1, 1, 0),
                "next_value": ("nv_ring", 0, None, 0)}[name]

    def field_shape(self, name):
        n = self.n_agents
        return {"state": (n, self.obs_dim), "next_state": (n, self.obs_dim), "last_hid": (n, self.hid_dim),
                "hid": (n, self.hid_dim), "action": (n, self.act_dim), "reward": (n,), "done": (), "last_step": (),
                "next_value": (n, 1)}[name]

    STORED = ("state", "action", "reward", "next_state", "done", "last_step", "last_hid", "hid")

    def gather(self, plan, slot, td=None):
        """Refresh static batch tensors from the window starting at global slot ``slot``: ONE launch of
        flexnet_gather_rows (include/flexnet.h).  ``plan`` = [(ring name, first column, width, row offset, rows, dst)],
        dst a contiguous [rows, width] fp32 tensor.  ``td`` = (reward tensor of the plan, FlexTdLossArgs): the value loss's
        reward-statistics pass rides in the same launch (flexnet_gather_rows_td; nets.offer_td_stats)."""
        import ctypes as C
        from . import _lib
        jobs = []
        td_first = td_count = None
        for ring_name, col0, width, row_off, rows, dst in plan:
            if td is not None and dst is td[0]:
                td_first = len(jobs)
            if ring_name == "stack_ring":                # read in place (enable_stacked_ring): only the window's first row moves
                cell = dst[0]
                self.expand_stacked()
                if rows + row_off > self.stack_tail:
                    raise ValueError("window longer than the stacked ring's mirrored tail")
                p = slot % self.stack_rows               # (views further into the window shift the ring's BASE, not the cell)
                jobs.append((self.slot_table.data_ptr() + 8 * p, cell.data_ptr(), 1, 2, 2))
                continue
            if ring_name == "row_ring":                  # stacked observations out of the row ring: a launch of its own kind
                self.stacked_obs(slot + row_off, rows, out=dst)
                continue
            ring = getattr(self, ring_name)
            stride = ring.shape[2]
            width = stride if width is None else width
            base, out, done_rows = ring.data_ptr(), dst.data_ptr(), 0
            for p, c in self.segments(slot + row_off, rows):
                jobs.append((base + 4 * (p * stride + col0), out + 4 * done_rows * width, c, width, stride))
                done_rows += c
            if td is not None and dst is td[0]:
                td_count = len(jobs) - td_first
        # a window that wraps the ring's seam splits every field in two: up to 2 x 8 stored fields = 16 jobs against the
        # launch's FLEXNET_GATHER_MAX_JOBS (12) — the rest goes out as a second launch instead of an intermittent error
        stream = C.c_void_p(th.cuda.current_stream().cuda_stream)
        for lo in range(0, len(jobs), _lib.FLEXNET_GATHER_MAX_JOBS):
            a = _lib.FlexGatherArgs()
            chunk = jobs[lo:lo + _lib.FLEXNET_GATHER_MAX_JOBS]
            for j, (src, dst_p, c, width, stride) in enumerate(chunk):
                a.src[j], a.dst[j] = src, dst_p
                a.rows[j], a.width[j], a.src_stride[j], a.dst_stride[j] = c, width, stride, width
            a.n_jobs = len(chunk)
            if td_count and lo <= td_first and td_first + td_count <= lo + len(chunk):
                _lib.check(_lib.load().flexnet_gather_rows_td(C.byref(a), td_first - lo, td_count, C.byref(td[1]), stream),
                           "flexnet_gather_rows_td")
                td_count = 0
            else:
                _lib.check(_lib.load().flexnet_gather_rows(C.byref(a), stream), "flexnet_gather_rows")
        if td is not None and td_count != 0:
            # (the reward's copies fell across two launches, or the plan does not gather the tensor: the pass on its own)
            _lib.check(_lib.load().flexnet_td_stats(C.byref(td[1]), stream), "flexnet_td_stats")

    def window_refresh_args(self, plan, start_cell, td=None):
        """``plan`` as FlexWindowRefreshArgs with the window's first slot read from the device cell ``start_cell`` (int64, a
        global slot number as sample_slot returns it) — the refresh a HIP graph can hold (trainer: one graph per update
        event).  Returns (args, FlexTdLossArgs or None), or None when the plan holds something this form does not cover (a
        window expanded out of the row ring, rings of different lengths, more jobs than the launch takes)."""
        from . import _lib
        a = _lib.FlexWindowRefreshArgs()
        a.start, a.ring_rows = start_cell.data_ptr(), self.slabs * self.n_envs
        nj = nc = 0
        td_args = None
        for ring_name, col0, width, row_off, rows, dst in plan:
            if ring_name == "stack_ring":
                if nc >= _lib.FLEXNET_WINDOW_MAX_CELLS or rows + row_off > self.stack_tail or self.stack_rows != a.ring_rows:
                    return None
                a.cell[nc], a.cell_mod[nc] = dst[0].data_ptr(), self.stack_rows
                nc += 1
                continue
            if ring_name == "row_ring" or nj >= _lib.FLEXNET_WINDOW_MAX_JOBS:
                return None
            ring = getattr(self, ring_name)
            if ring.shape[0] * ring.shape[1] != a.ring_rows:
                return None
            stride = ring.shape[2]
            width = stride if width is None else width
            a.base[nj], a.dst[nj] = ring.data_ptr() + 4 * col0, dst.data_ptr()
            a.rows[nj], a.row_off[nj], a.width[nj], a.src_stride[nj] = rows, row_off, width, stride
            if td is not None and dst is td[0]:
                a.reward_job, td_args = nj, td[1]
            nj += 1
        if td is not None and td_args is None:
            return None
        a.n_jobs, a.n_cells = nj, nc
        return a, td_args

    def window_refresh(self, args):
        """Launch a refresh prepared by window_refresh_args on the current stream (capturable)."""
        import ctypes as C
        from . import _lib
        a, td_args = args
        _lib.check(_lib.load().flexnet_window_refresh(C.byref(a), C.byref(td_args) if td_args is not None else None,
                                                      C.c_void_p(th.cuda.current_stream().cuda_stream)), "flexnet_window_refresh")

    def scatter(self, ring_name, src, slot, rows):
        """The reverse of one gather job: rows of the contiguous [rows, width] tensor ``src`` into the ring's slots of the
        global slot range [slot, slot + rows) (two pieces at the seam): one launch of flexnet_gather_rows."""
        import ctypes as C
        from . import _lib
        ring = getattr(self, ring_name)
        stride = ring.shape[2]
        if not (src.is_contiguous() and src.dtype == th.float32 and src.numel() == rows * stride):
            raise ValueError("scatter: src must be a contiguous fp32 [rows, row width of the ring] tensor")
        a = _lib.FlexGatherArgs()
        done_rows = 0
        for j, (p, c) in enumerate(self.segments(slot, rows)):
            a.src[j], a.dst[j] = src.data_ptr() + 4 * done_rows * stride, ring.data_ptr() + 4 * p * stride
            a.rows[j], a.width[j], a.src_stride[j], a.dst_stride[j] = c, stride, stride, stride
            done_rows += c
            a.n_jobs = j + 1
        _lib.check(_lib.load().flexnet_gather_rows(C.byref(a), C.c_void_p(th.cuda.current_stream().cuda_stream)),
                   "flexnet_gather_rows")

    def _logical_to_slot(self, index):
        """Global slot of logical transition ``index`` (0 = oldest), skipping gaps."""
        N = self.n_envs
        for a, b in self._runs():
            cnt = (b - a) * N
            if index < cnt:
                return a * N + index
            index -= cnt
        raise IndexError("transition index out of range")

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
        if self.slab_mode:
            raise RuntimeError("this replay buffer is in slab mode (written by flexnet_rollout_pack); add_batch is the "
                               "field-by-field mode of the eager rollout")
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
        if self.slab_mode:
            self.first, self.gaps = self.k, []

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
        if self.slab_mode:
            slot = self._logical_to_slot(start)
            if self._logical_to_slot(start + batch_size - 1) != slot + batch_size - 1:
                raise ValueError("window spans a gap of the slab ring")
            return self.slab_window(slot, batch_size)
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
        if self.slab_mode:
            return self.slab_window(self.sample_slot(batch_size), batch_size)
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
