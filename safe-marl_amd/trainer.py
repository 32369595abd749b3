"""Runner for the MADDPG path with the reference's trainer surface (utils/trainer.py:10-138), i.e. what
train_agent.py:107-146 and model.py:40-71 touch: ``PGTrainer(args, model_cls, env, logger)``, ``run``,
``logging``, ``print_info``, ``behaviour_net``, ``replay_buffer``, ``steps``, ``episodes`` and the two
``*_replay_process`` hooks.

Built for device-resident data instead of per-step Python objects:
  * the replay memory is ``replay_buffer.TransReplayBuffer`` in HBM and hands out tensor windows — no
    re-columnisation of a list of transitions per sub-update (trainer.py:66-74);
  * a sub-update evaluates only the loss it steps on (the reference evaluates both, trainer.py:84,101);
  * statistics stay on the device until the end of an episode (no ``.item()`` per sub-update);
  * with more than one rank the gradients travel as ONE flat bucket through RCCL, averaged BEFORE the
    grad-norm clip;
  * with a vectorised env a batch is ``batch_size * batch_scale`` consecutive replay slots.
"""
from __future__ import annotations

import logging
import os

import numpy as np
import torch as th
from torch.optim import RMSprop

from . import dist as fdist
from .optim import clip_and_step
from .replay_buffer import TransReplayBuffer
from .util import graph_capture, normal_entropy

train_logger = logging.getLogger("TrainLogger")

_RMSPROP = dict(alpha=0.99, eps=1e-5)       # trainer.py:34-35


class PGTrainer(object):
    def __init__(self, args, model, env, logger, batch_scale=None, replay_capacity=None, graph_rollout=True,
                 graph_updates=True, sync_reward_bn=None):
        if args.episodic:
            raise NotImplementedError("episodic replay is outside the MADDPG hot path (default.yaml:9)")
        self.args, self.env, self.logger = args, env, logger
        self.episodic = False
        self.device = th.device("cuda" if th.cuda.is_available() and args.cuda else "cpu")
        self.steps = self.episodes = 0
        self.graph_rollout = graph_rollout      # vectorised envs: replay each rollout step as one HIP graph
        self.graph_updates = graph_updates      # ... and each sub-update (packed replay on the GPU, one rank)
        self._update_graphs = {}
        self._update_graphs_alt = {}            # second static batch per kind for the pipelined update event
        self._side_stream = None
        self._entr_terms = {}
        # replay_event can gather window j + 1 on a side stream while sub-update j runs.  Measured (bench.py training legs,
        # A/B on one box): the overlapped gather fights the replay for HBM — its own duration stretches from 49 to ~140 us,
        # the replay's first GEMM from 31 to ~48 us — and the loop is 4-8 % SLOWER than gather-then-replay, so it is off
        # unless asked for (FLEX_PIPELINE_UPDATES=1); bit-identical either way (tests/test_update_graph_gpu.py)
        self.pipeline_updates = os.environ.get("FLEX_PIPELINE_UPDATES") == "1"
        # replay_event: bootstrap values filed once per update event where the windows overlap enough (FLEX_BOOTSTRAP_CACHE=0: never)
        self.cache_bootstrap = os.environ.get("FLEX_BOOTSTRAP_CACHE", "1") != "0"
        self._bootstrap_graphs = {}                    # pass size -> graph of MADDPG.bootstrap_values on a static batch
        self._cached_graphs = {}
        self._cached_ready = False
        # the value sub-updates of an update event as ONE graph (replay_event; FLEX_EVENT_GRAPH=0: one graph launch each)
        self.event_graphs = os.environ.get("FLEX_EVENT_GRAPH", "1") != "0"
        self._event_graphs = {}                        # (kind, count) -> graph of `count` sub-updates, or False (not possible)
        self.event_graph_replays = 0
        self.bootstrap_cached_events = 0
        # more than one rank: the sub-update region is split at the exchange step — graph A (loss, gradients into one flat
        # bucket), eager all-reduce, graph B (clip, RMSprop) — the form that has run on hardware (gloo ranks on one GPU, RCCL
        # at world size 1).  FLEX_ALLREDUCE_IN_GRAPH=1 opts in to ONE graph with the all-reduce captured inside it (RCCL
        # collectives can be stream-captured: ProcessGroupNCCL joins its stream to the capture); it stays opt-in until a
        # multi-GPU RCCL run of tests/test_dist_gpu.py exists, and falls back to the split form if the capture
        # fails
        self.entr = args.entr
        self.world = fdist.world_size()
        # Reward BatchNorm over ALL ranks' batches ("or all-reduce 2 x 5 moments"): off by default (per-rank
        # statistics, the documented deviation); on (argument or FLEX_SYNC_REWARD_BN=1) every reward normalisation sums its
        # moments over the ranks first — one more 8 KB all-reduce per sub-update, inside the update graph where the gradient
        # all-reduce is (nccl), eager sub-updates otherwise
        if sync_reward_bn is None:
            sync_reward_bn = os.environ.get("FLEX_SYNC_REWARD_BN") == "1"
        self.sync_reward_bn = bool(sync_reward_bn) and self.world > 1
        # cross-rank reward statistics put an all-reduce INSIDE the loss (graph A): on nccl they imply the in-graph form
        # (left independent, sync_reward_bn on nccl without FLEX_ALLREDUCE_IN_GRAPH silently dropped the whole run
        # to eager sub-updates); on gloo nothing can be captured and the sub-updates run eagerly — on EVERY rank (all_agree)
        self.allreduce_in_graph = ((os.environ.get("FLEX_ALLREDUCE_IN_GRAPH", "0") == "1" or self.sync_reward_bn)
                                   and self.world > 1 and fdist.backend() == "nccl")

        # behaviour net (+ target replica), trainer.py:16-28: SAFEMADDPG also receives the env
        ctor_args = (args, env) if args.alg == "safemaddpg" else (args,)
        if args.target:
            self.behaviour_net = model(*ctor_args, model(*ctor_args).to(self.device)).to(self.device)
        else:
            self.behaviour_net = model(*ctor_args).to(self.device)
        if self.world > 1:                      # identical replicas: rank 0's weights everywhere
            fdist.broadcast_module(self.behaviour_net)
        net = self.behaviour_net
        from .nets import set_reward_bn_sync
        for m in (net, getattr(net, "target_net", None)):          # the flag lives on THIS trainer's BatchNorm modules
            set_reward_bn_sync(getattr(m, "batchnorm", None), self.sync_reward_bn)
        cap = dict(capturable=True) if self.device.type == "cuda" else {}       # optimiser steps inside HIP graphs
        self.policy_optimizer = RMSprop(net.policy_dicts.parameters(), lr=args.policy_lrate, **_RMSPROP, **cap)
        self.value_optimizer = RMSprop(net.value_dicts.parameters(), lr=args.value_lrate, **_RMSPROP, **cap)
        self.init_action = th.zeros(1, args.agent_num, args.action_dim, device=self.device)

        # replay memory sized for the number of environments feeding it
        n_envs = getattr(env, "n_envs", 1)
        if batch_scale is None:
            batch_scale = 1 if n_envs == 1 else max(1, n_envs // 4)
        self.batch_scale = batch_scale
        if args.replay:
            if replay_capacity is None:
                replay_capacity = int(args.replay_buffer_size) * max(1, min(n_envs, 64))
            self.replay_buffer = TransReplayBuffer(replay_capacity, device=self.device)

    # ---- sampling ----------------------------------------------------------------------------
    def effective_batch_size(self):
        return self.args.batch_size * self.batch_scale

    def get_loss(self, batch, need="both"):
        return self.behaviour_net.get_loss(batch, need=need)

    def policy_replay_process(self, stat):      # model.py:50
        if not self._graphed_sub_update("policy", stat):
            self._sub_update("policy", stat, self.replay_buffer.get_batch_tensors(self.effective_batch_size()))

    def value_replay_process(self, stat):       # model.py:48
        if not self._graphed_sub_update("value", stat):
            self._sub_update("value", stat, self.replay_buffer.get_batch_tensors(self.effective_batch_size()))

    # ---- a sub-update as one HIP graph -----------------------------------------------------------------------
    def _ensure_graph(self, which, slot=0):
        """The captured sub-update of kind ``which`` on static batch ``slot`` (0, or 1 for the double-buffered pipeline of
        replay_event), capturing it on first use; None when this configuration does not qualify for graphed updates."""
        buf = self.replay_buffer
        if not (self.graph_updates and self.device.type == "cuda" and getattr(buf, "slab_mode", False)):
            return None
        if not getattr(self.behaviour_net, "graph_safe_updates", False):
            # Only models whose gradient path is free of PyTorch's multi-block reductions are replayed as graphs: with
            # this PyTorch-ROCm build such a reduction (global semaphore + memset) captured into a HIP graph can come back
            # stale or partial on replay.  MADDPG / SAFEMADDPG reduce with this project's fixed-order
            # kernels and are checked against eager updates at full size; so do MATD3 / IDDPG since round 2 (a model outside
            # this package declares graph_safe_updates only if FLEX_GRAPH_AUDIT=1 accepts both of its sub-update bodies).
            return None
        bs = self.effective_batch_size()
        store = self._update_graphs if slot == 0 else self._update_graphs_alt
        if which == "value_cached":                   # (kept apart: _update_graphs lists the sub-updates of model.py:47-50)
            store = self._cached_graphs
            which_key = (which, slot)
        else:
            which_key = which
        g = store.get(which_key)
        if g is None or g["bs"] != bs or g["buf"] is not buf or g.get("ring_gen") != getattr(buf, "stack_gen", None):
            exc = None
            try:
                g = self._capture_sub_update(which, bs)
            except Exception as e:                    # capture not possible here: stay eager from now on
                g, exc = None, e
            # every rank reaches this point at the same sub-update (same schedule): the fallback is taken by ALL of them or by
            # none — a rank replaying graphs beside a rank stepping eagerly would still match collective for collective today,
            # but only by accident of the two forms' call sequences
            if not fdist.all_agree(g is not None, self.device):
                import warnings
                why = exc if exc is not None else "another rank's capture failed"
                if which == "value_cached":           # ... or just without the filed bootstrap values
                    warnings.warn(f"capture of the value sub-update on filed bootstrap values failed ({why}); computing them per sub-update")
                    self.cache_bootstrap = False
                    return None
                warnings.warn(f"sub-update graph capture failed ({why}); using eager sub-updates")
                self.graph_updates = False
                return None
            g["ring_gen"] = getattr(buf, "stack_gen", None)
            store[which_key] = g
        return g

    def _replay(self, g, stat):
        g["graph"].replay()
        if g["flat"] is not None and g["apply"] is None:          # the all-reduce captured inside the graph ran once more
            fdist.note_allreduce(g["flat"].numel() * g["flat"].element_size())
        if g["apply"] is not None:
            # more than one rank: the captured region is split at the exchange step — graph A (losses, backward, gradients
            # into ONE static flat bucket), the all-reduce of that bucket through RCCL (eager: one ncclAllReduce of
            # <= 208 KB), graph B (scale by 1/world, gradient clip, RMSprop).  SURVEY.md §8(e): the clip comes AFTER.
            fdist.allreduce_flat(g["flat"])
            g["apply"].replay()
        stat.update(g["stat"])

    def _graphed_sub_update(self, which, stat):
        """A sub-update is ~120 kernel launches that take the host longer to issue (1.6 ms) than the GPU to run; with
        the replay in slab mode the sampled window is gathered into a static batch (ONE launch, include/flexnet.h:
        flexnet_gather_rows) and the whole step — losses, backward, gradient clip, RMSprop — is replayed as one HIP graph.
        Returns False when this configuration does not qualify (the caller then runs the eager step)."""
        g = self._ensure_graph(which)
        if g is None:
            return False
        buf = self.replay_buffer
        buf.gather(g["plan"], buf.sample_slot(g["bs"]), td=g.get("td"))       # only the ring columns this sub-update reads
        self._replay(g, stat)
        return True

    def replay_event(self, stat, n_value, n_policy):
        """One update event of model.py:47-50 — ``n_value`` value sub-updates, then ``n_policy`` policy sub-updates.

        Round 3, bootstrap values filed once per event: Q'(s', pi(s')) is 43 % of a value sub-update, and the networks behind
        it (behaviour or target policy, target critic) do not change before the event's policy sub-update.  The windows of the
        value sub-updates are drawn up front (same NumPy draws, same order), the UNION of their transitions is valued in
        passes of one batch into the ring's ``nv_ring`` (a graph of MADDPG.bootstrap_values, the very lines the loss runs), and
        the value sub-updates read their window's values from there ("value_cached" graphs).  At the reference's sample reuse
        (32 slabs per window in a ring of 192) ten windows overlap to ~6 passes; at batch_scale = N / 4 they hardly overlap, the
        passes would outnumber what they save, and the event runs as before.  Same values either way — same kernels on the
        same rows — so the choice is made per event (tests/test_update_graph_gpu.py).

        With FLEX_PIPELINE_UPDATES=1 the event is also software-pipelined: the window of sub-update j + 1 is gathered into
        the OTHER static batch of its kind on a side stream while the graph of sub-update j runs (bit-identical; measured
        4-8 % slower than gather-then-replay, hence off by default)."""
        n_value, n_policy = int(n_value), int(n_policy)
        kinds = ["value"] * n_value + ["policy"] * n_policy
        if not kinds or self._ensure_graph(kinds[0]) is None:
            for which in kinds:
                (self.value_replay_process if which == "value" else self.policy_replay_process)(stat)
            return
        buf = self.replay_buffer
        bs_all = self.effective_batch_size()
        starts, chunks, boot = {}, [], None
        net = self.behaviour_net
        # every graph an event may replay exists after the FIRST event (a look-up afterwards): a capture at first use would
        # land in the middle of somebody's timed region
        self._ensure_event_graph(kinds[0], n_value)
        # (Passes of ONE size, the sub-update's batch: quarter passes for what an interval leaves over were tried — 12.4 -> 12.0
        # ms per event at the reference's reuse — and given up: at another row count the library picks another first-layer
        # GEMM kernel, the values differ in the seventh digit, and the event is no longer bit-identical to the plain one.)
        eligible = (self.cache_bootstrap and n_value >= 3 and getattr(buf, "nv_ring", None) is not None
                    and "value_cached" in (getattr(net, "update_fields", None) or {}) and hasattr(net, "bootstrap_values")
                    and getattr(net, "bootstrap_cacheable", False) and getattr(net, "target_net", None) is not None)
        if eligible and not self._cached_ready:
            # The cached form is captured at the FIRST eligible event, whether or not its windows overlap enough: a capture
            # at first use would land in the middle of somebody's timed region (what round 2's config-4 figure suffered
            # from), and with more than one rank a capture's warm-up steps all-reduce, so every rank must capture at the
            # same point.
            self._cached_ready = True
            ok = self._ensure_graph("value_cached") is not None and self._ensure_bootstrap(bs_all) is not None
            if ok and self.pipeline_updates:
                ok = self._ensure_graph("value_cached", 1) is not None
            if not ok:
                self.cache_bootstrap = eligible = False
            else:
                self._ensure_event_graph("value_cached", n_value)       # (the event's one-graph form, captured at the same point)
        elif eligible:
            # Whether THIS event takes the cached form is decided below from this rank's OWN window draws (how much they overlap);
            # whether the cached form's graphs are current must not be.  A graph goes stale when the replay's stacked ring is
            # re-allocated (a later capture asking for a longer tail: ``stack_gen``), and recapturing it runs warm-up steps that
            # all-reduce and ends in an agreement — left to the first event that CHOSE the cached form, one rank recaptured while
            # another went on with its plain sub-updates, and their collectives no longer matched (found by the two-rank bench
            # rehearsal, round 5: `52100 vs 4` in gloo's pair).  So every rank brings them up to date here, at the same event;
            # afterwards the look-ups below capture nothing.  (One rank: the same recapture, a few lines earlier.)
            ok = self._ensure_graph("value_cached") is not None and self._ensure_bootstrap(bs_all) is not None
            if ok and self.pipeline_updates:
                ok = self._ensure_graph("value_cached", 1) is not None
            if not ok:
                self.cache_bootstrap = eligible = False
        # ... and the PLAIN form's graph once more: the cached form's capture above may have re-allocated the stacked ring (SAFEMADDPG:
        # the plain value sub-update does not read its window in place, the cached one does and creates the ring), which leaves the
        # graph captured a few lines up stale.  Recaptured lazily, that happened on the ranks whose event went on in the plain form
        # and not on those that took the cached one — the two-rank bench rehearsal's `52100 vs 4` (collective log: rank 1
        # in _replay_event_plain -> _ensure_graph -> capture, rank 0 in _replay).  A look-up when nothing moved.
        if self._ensure_graph(kinds[0]) is None or (self.pipeline_updates and self.world > 1 and self._ensure_graph(kinds[0], 1) is None):
            for which in kinds:
                (self.value_replay_process if which == "value" else self.policy_replay_process)(stat)
            return
        self._ensure_event_graph(kinds[0], n_value)             # (its base graph may just have been recaptured)
        if eligible:
            vstarts = [buf.sample_slot(bs_all) for _ in range(n_value)]
            starts = dict(enumerate(vstarts))
            N = buf.n_envs
            chunks = self.bootstrap_chunks(vstarts, bs_all, [(a * N, b * N) for a, b in buf._runs()])
            # a pass costs about what a value sub-update saves: the cached form has to save at least two of them to be chosen
            choose = len(chunks) + 2 < n_value
            if getattr(self, "force_bootstrap_choice", None) is not None:      # (tests: ranks made to choose differently)
                choose = bool(self.force_bootstrap_choice)
            if choose:
                boot = {size: self._ensure_bootstrap(size) for size in {size for _, size in chunks}}
                if (any(g is None for g in boot.values()) or self._ensure_graph("value_cached") is None or
                        (self.pipeline_updates and self._ensure_graph("value_cached", 1) is None)):
                    boot = None
        if boot is not None:
            kinds = ["value_cached"] * n_value + ["policy"] * n_policy
            for c, size in chunks:                      # the union of the value windows, valued once
                g = boot[size]
                buf.gather(g["plan"], c)
                g["graph"].replay()
                buf.scatter("nv_ring", g["nv"], c, size)
            self.bootstrap_cached_events += 1
        if not self.pipeline_updates or len(kinds) < 2:
            return self._replay_event_plain(stat, kinds, starts)
        seen, graphs = {}, []
        for which in kinds:                             # every (kind, slot) graph exists before the pipeline starts
            slot = seen.get(which, 0) % 2
            seen[which] = seen.get(which, 0) + 1
            g = self._ensure_graph(which, slot)
            if g is None:                               # capture failed half-way: plain calls for the whole event
                return self._replay_event_plain(stat, kinds, starts)
            graphs.append(g)
        main = th.cuda.current_stream()
        if self._side_stream is None:
            self._side_stream = th.cuda.Stream()
        side = self._side_stream
        side.wait_stream(main)                          # the ring writes of the rollout so far (and the filed values)

        def launch_gather(j):
            g = graphs[j]
            # drawn in sub-update order, like the one-at-a-time calls (the value windows possibly up front, above)
            slot_start = starts[j] if j in starts else buf.sample_slot(g["bs"])
            with th.cuda.stream(side):
                if g.get("free") is not None:
                    side.wait_event(g["free"])          # the last replay that read this static batch has finished
                buf.gather(g["plan"], slot_start, td=g.get("td"))
                ev = th.cuda.Event()
                ev.record(side)
            return ev

        ready = launch_gather(0)
        for j, g in enumerate(graphs):
            nxt = launch_gather(j + 1) if j + 1 < len(graphs) else None
            main.wait_event(ready)
            self._replay(g, stat)
            g["free"] = th.cuda.Event()
            g["free"].record(main)
            ready = nxt

    def _ensure_event_graph(self, kind, count):
        """``count`` consecutive sub-updates of ``kind`` ("value" / "value_cached") as ONE HIP graph: each refreshes the static
        batch of the kind's single-step graph from ITS cell of a device array of window starts (flexnet_window_refresh, the
        statistics rider included) and runs the very body that graph holds.  A graph launch ends with ~8 us of hand-over before
        the next thing on the stream starts; an update event of model.py:47-50 paid that ten times.  One rank only (the split
        form's eager all-reduce sits between two graphs), not with the side-stream refresh.  None: run them one at a time."""
        if not self.event_graphs or self.world > 1 or self.pipeline_updates or count < 2:
            return None
        g = self._ensure_graph(kind)
        if g is None:
            return None
        key = (kind, count)
        eg = self._event_graphs.get(key)
        if eg is not None and (eg is False or eg["base"] is g):
            return eg or None
        buf = self.replay_buffer
        starts = th.zeros(count, dtype=th.int64, device=self.device)
        refresh = [buf.window_refresh_args(g["plan"], starts[j:j + 1], g.get("td")) for j in range(count)]
        eg = False
        if all(r is not None for r in refresh):
            graph, out = th.cuda.CUDAGraph(), {}
            self.behaviour_net.bootstrap_from_batch = kind == "value_cached"
            try:
                # (no warm-up of its own: the single-step graph's capture has run these launches on this batch already, and a
                #  capture executes nothing — weights, optimiser state and statistics are untouched)
                # the refresh for sub-update j + 1 rides in the optimiser launches that end sub-update j (optim.clip_and_step:
                # flexnet_clip_rmsprop_refresh; FLEX_REFRESH_RIDER=0: a launch of its own)
                ride = os.environ.get("FLEX_REFRESH_RIDER", "1") != "0"
                with graph_capture(graph):
                    buf.window_refresh(refresh[0])
                    for j in range(count):
                        nxt = refresh[j + 1] if j + 1 < count else None
                        self._next_refresh = nxt if ride else None
                        self._sub_update("value", out, g["batch"], fresh_leaves=True)
                        if nxt is not None and (not ride or self._next_refresh is not None):
                            buf.window_refresh(nxt)              # (not taken along by the step)
                        self._next_refresh = None
                eg = dict(graph=graph, stat=out, starts=starts, base=g, refresh=refresh,
                          expands=any(p[0] == "stack_ring" for p in g["plan"]))
            except Exception as exc:
                import warnings
                warnings.warn(f"update-event graph capture failed ({exc}); one graph launch per sub-update")
                th.cuda.synchronize()
            finally:
                self.behaviour_net.bootstrap_from_batch = False
                self._next_refresh = None
        self._event_graphs[key] = eg
        return eg or None

    def _replay_event_plain(self, stat, kinds, starts):
        """Gather, then replay, one sub-update at a time — runs of value sub-updates as one graph where that form exists
        (_ensure_event_graph); ``starts``: windows already drawn (index in ``kinds`` -> slot)."""
        buf = self.replay_buffer
        done_upto = 0
        for j, which in enumerate(kinds):
            if j < done_upto:
                continue
            if which in ("value", "value_cached"):
                run = 1
                while j + run < len(kinds) and kinds[j + run] == which:
                    run += 1
                eg = self._ensure_event_graph(which, run)
                if eg is not None:
                    bs = eg["base"]["bs"]
                    # drawn in sub-update order, like the one-at-a-time calls (nothing else draws from the NumPy stream in between)
                    slots = [starts[i] if i in starts else buf.sample_slot(bs) for i in range(j, j + run)]
                    if eg["expands"]:
                        buf.expand_stacked()
                    eg["starts"].copy_(th.tensor(slots, dtype=th.int64))       # (pageable source: staged before the call returns)
                    eg["graph"].replay()
                    stat.update(eg["stat"])
                    self.event_graph_replays += 1
                    done_upto = j + run
                    continue
            g = self._ensure_graph(which)
            if g is None:                               # (graphs went away mid-event: the eager step on the same window)
                need = "value" if which == "value_cached" else which
                slot = starts[j] if j in starts else buf.sample_slot(self.effective_batch_size())
                self._sub_update(need, stat, buf.slab_window(slot, self.effective_batch_size()))
                continue
            buf.gather(g["plan"], starts[j] if j in starts else buf.sample_slot(g["bs"]), td=g.get("td"))
            self._replay(g, stat)

    @staticmethod
    def bootstrap_chunks(starts, bs, runs=()):
        """Passes of ``bs`` consecutive transitions that cover the union of the windows [s, s + bs): [(start, bs)].  The windows
        are merged into intervals; two neighbouring intervals inside the same run of complete transitions (``runs``: [(lo,
        hi)] global slot ranges, TransReplayBuffer._runs) are joined across the gap between them when that takes fewer
        passes; an interval is cut into passes from its left end, the last one flush with its right end."""
        ivals = sorted((int(s), int(s) + bs) for s in starts)
        merged = []
        a, b = ivals[0]
        for x, y in ivals[1:]:
            if x <= b:
                b = max(b, y)
            else:
                merged.append((a, b))
                a, b = x, y
        merged.append((a, b))

        def passes(lo, hi):
            return -(-(hi - lo) // bs)

        def same_run(lo, hi):
            return any(r0 <= lo and hi <= r1 for r0, r1 in runs)

        joined = [merged[0]]
        for x, y in merged[1:]:
            a, b = joined[-1]
            if same_run(a, y) and passes(a, y) < passes(a, b) + passes(x, y):
                joined[-1] = (a, y)
            else:
                joined.append((x, y))
        chunks = []
        for a, b in joined:
            c = a
            while c + bs <= b:
                chunks.append((c, bs))
                c += bs
            if c < b:
                chunks.append((b - bs, bs))
        return chunks

    def _ensure_bootstrap(self, bs):
        g = self._bootstrap_graphs.get(bs)
        if g is None or g["buf"] is not self.replay_buffer or g.get("ring_gen") != getattr(self.replay_buffer, "stack_gen", None):
            exc = None
            try:
                g = self._capture_bootstrap(bs)
            except Exception as e:
                g, exc = None, e
            # (captures are attempted at the same event on every rank — replay_event — so this is reached by all of them or by
            #  none; a rank that gave up the cached form alone would skip the NEXT recapture of its graphs, collectives included)
            if not fdist.all_agree(g is not None, self.device):
                import warnings
                why = exc if exc is not None else "another rank's capture failed"
                warnings.warn(f"bootstrap-value graph capture failed ({why}); value sub-updates compute their own")
                self.cache_bootstrap = False
                return None
            g["ring_gen"] = getattr(self.replay_buffer, "stack_gen", None)
            self._bootstrap_graphs[bs] = g
        return g

    def _static_batch(self, which, bs):
        """Static tensors a captured sub-update reads, and the gather plan that refreshes them.  Every observation sits
        in the ring once, so when a loss reads both ``state`` and ``next_state`` they are two views — N rows apart — of
        ONE gathered block of bs + N observation rows (9/16 of the bytes of two separate copies at the default batch)."""
        buf = self.replay_buffer
        N = buf.n_envs
        names = (getattr(self.behaviour_net, "update_fields", None) or {}).get(which) or buf.STORED
        fields, plan = {}, []
        dev = self.device

        def block(rows, width):
            return th.zeros(rows, width, dtype=th.float32, device=dev)

        in_place = (which == "value_cached" and getattr(buf, "row_mode", False) and dev.type == "cuda"
                    and os.environ.get("FLEX_STACKED_RING", "1") != "0" and "state" in names and "next_state" not in names
                    and getattr(self.behaviour_net, "reads_state_in_place", lambda _bs: False)(bs))
        if in_place:
            # the value sub-update on filed bootstrap values reads its observations IN PLACE from the replay's stacked ring
            # (nets.RING_VIEWS): no gather of the window, `state` is a NaN placeholder that only ring-aware kernels may touch
            from . import nets
            w = buf.n_agents * buf.obs_dim
            buf.enable_stacked_ring(bs + N)
            ph = th.full((bs, w), float("nan"), dtype=th.float32, device=dev)
            cell = th.zeros(1, dtype=th.int64, device=dev)
            nets.register_ring_view(ph, buf.stack_ring, cell)
            plan.append(("stack_ring", 0, None, 0, bs, (cell, ph)))
            fields["state"] = ph.view((bs,) + buf.field_shape("state"))
        in_place_next = (which == "bootstrap" and getattr(buf, "row_mode", False) and dev.type == "cuda"
                         and os.environ.get("FLEX_STACKED_RING", "1") != "0" and "next_state" in names and "state" not in names
                         and not getattr(self, "_no_ring_bootstrap", False)
                         and getattr(self.behaviour_net, "reads_next_state_in_place", lambda _bs: False)(bs))
        if in_place_next:                             # the same for the passes that file the bootstrap values: next_state = slot + N
            from . import nets
            w = buf.n_agents * buf.obs_dim
            buf.enable_stacked_ring(bs + N)
            ph = th.full((bs, w), float("nan"), dtype=th.float32, device=dev)
            cell = th.zeros(1, dtype=th.int64, device=dev)
            nets.register_ring_view(ph, buf.stack_ring[N:], cell)
            keep = [ph]
            fields["next_state"] = ph.view((bs,) + buf.field_shape("next_state"))
            keep += self._hid_in_place(buf, names, fields, cell, bs)
            plan.append(("stack_ring", 0, None, N, bs, (cell, keep)))
        in_place_both = (which == "value" and getattr(buf, "row_mode", False) and dev.type == "cuda"
                         and os.environ.get("FLEX_STACKED_RING", "1") != "0" and "state" in names and "next_state" in names
                         and getattr(self.behaviour_net, "reads_state_in_place", lambda _bs: False)(bs)
                         and getattr(self.behaviour_net, "reads_next_state_in_place", lambda _bs: False)(bs))
        if in_place_both:                             # the plain value sub-update: both views of the window, N rows apart
            from . import nets
            w = buf.n_agents * buf.obs_dim
            buf.enable_stacked_ring(bs + N)
            cell = th.zeros(1, dtype=th.int64, device=dev)           # ONE cell: the window's first row; next_state = N rows on
            keep = []
            for name, off in (("state", 0), ("next_state", N)):
                ph = th.full((bs, w), float("nan"), dtype=th.float32, device=dev)
                nets.register_ring_view(ph, buf.stack_ring[off:], cell)
                keep.append(ph)
                fields[name] = ph.view((bs,) + buf.field_shape(name))
            keep += self._hid_in_place(buf, names, fields, cell, bs)
            plan.append(("stack_ring", 0, None, N, bs, (cell, keep)))
        if "state" in names and "next_state" in names and not in_place_both:
            w = buf.n_agents * buf.obs_dim
            win = block(bs + N, w)
            plan.append((buf.obs_source_ring, 0, None, 0, bs + N, win))
            fields["state"] = win[:bs].view((bs,) + buf.field_shape("state"))
            fields["next_state"] = win[N:N + bs].view((bs,) + buf.field_shape("next_state"))
        for k in names:
            if k in fields:
                continue
            ring, col0, width, slab_off = buf.field_source(k)
            shape = buf.field_shape(k)
            w = int(np.prod(shape)) if len(shape) else 1
            t = block(bs, w)
            plan.append((ring, col0, width, slab_off * N, bs, t))
            fields[k] = t.view((bs,) + shape)
        for k in buf.STORED:                          # never read by this loss: a broadcast zero of the right shape
            if k not in fields:
                shape = buf.field_shape(k)
                fields[k] = th.zeros((1,) + tuple(1 for _ in shape), device=dev).expand((bs,) + shape)
        for k, c in buf.consts.items():
            if k in fields:                           # (next_value: a constant of the ring, a gathered field of value_cached)
                continue
            shape = buf.const_shapes.get(k, ())
            fields[k] = th.full((1,) + tuple(1 for _ in shape), float(c), device=dev).expand((bs,) + tuple(shape))
            fields[k]._flex_const = float(c)
        return fields, plan

    @staticmethod
    def _hid_in_place(buf, names, fields, cell, bs):
        """``hid`` (the hidden state the policy starts from at next_state: slab + 1) read in place from the replay's
        hidden-state ring beside the observations of the same window — its only reader on these paths is the policy's fused
        inference pass (nets.fused_actor_forward).  Needs the window (and its + N view) inside the ring's mirrored tail."""
        from . import nets
        N = buf.n_envs
        if ("hid" not in names or "hid" in fields or os.environ.get("FLEX_HID_IN_PLACE", "1") == "0"
                or -(-(bs + N) // N) + 1 > getattr(buf, "hid_tail_slabs", 0)):
            return []
        w = buf.n_agents * buf.hid_dim
        flat = buf.hid_store.view(-1, w)
        ph = th.full((bs, w), float("nan"), dtype=th.float32, device=flat.device)
        nets.register_ring_view(ph, flat[N:], cell)
        fields["hid"] = ph.view((bs,) + buf.field_shape("hid"))
        return [ph]

    def _capture_bootstrap(self, bs):
        """The graph that files Q'(s', pi(s')) for ``bs`` consecutive transitions: next observations and hidden states of the
        window gathered into a static batch, MADDPG.bootstrap_values on it (the very lines the value loss runs), the result
        left in a static [bs, n] tensor that replay_event scatters into the ring's ``nv_ring``."""
        from .replay_buffer import Transition
        buf = self.replay_buffer
        net = self.behaviour_net
        fields, plan = self._static_batch("bootstrap", bs)
        buf.gather(plan, buf.warmup_slot(bs + buf.n_envs))
        batch = Transition(**fields)
        nv = th.zeros(bs, buf.n_agents, dtype=th.float32, device=self.device)

        def body():
            nv.copy_(net.bootstrap_values(batch.next_state, batch.action_avail, batch.hid))

        side = th.cuda.Stream()
        side.wait_stream(th.cuda.current_stream())
        with th.cuda.stream(side):
            for _ in range(2):
                body()
        th.cuda.current_stream().wait_stream(side)
        if any(p[0] == "stack_ring" for p in plan) and not bool(th.isfinite(nv).all()):
            # `next_state` is a NaN placeholder for ring-aware kernels (nets.RING_VIEWS); something else read it: this trainer
            # goes back to gathering the window
            import warnings
            warnings.warn("a consumer that cannot read the stacked-observation ring in place touched the bootstrap pass's "
                          "placeholder; gathering the window instead")
            self._no_ring_bootstrap = True
            return self._capture_bootstrap(bs)
        graph = th.cuda.CUDAGraph()
        with graph_capture(graph):
            body()
        return dict(graph=graph, plan=plan, bs=bs, buf=buf, batch=batch, nv=nv)

    def _capture_sub_update(self, which, bs):
        from .replay_buffer import Transition
        buf = self.replay_buffer
        # "value_cached": the value sub-update on bootstrap values filed by replay_event (MADDPG.bootstrap_from_batch)
        kind, cached = which, which == "value_cached"
        which = "value" if cached else which
        self.behaviour_net.bootstrap_from_batch = cached
        ar0, ag0 = fdist.STATS["allreduce_calls"], fdist.STATS["agreements"]
        try:
            return self._capture_sub_update_body(kind, which, bs)
        except Exception as exc:
            if self.world > 1:
                self._realign_after_failed_capture(kind, which, exc, ar0, ag0)
            raise
        finally:
            self.behaviour_net.bootstrap_from_batch = False

    def _realign_after_failed_capture(self, kind, which, exc, ar0, ag0):
        """More than one rank: a capture that fails on THIS rank must leave it at the same place in the sequence of collectives as
        the ranks whose capture went through — they ran two warm-up steps (one all-reduce of the gradient bucket each; a third
        under FLEX_GRAPH_AUDIT) and, in the opt-in one-graph form, one agreement on that form — or the agreement that follows
        (_ensure_graph: everybody falls back together) meets a peer's gradient bucket instead of a peer's flag.  The missing
        all-reduces are issued on a zero bucket of the right size (the peers' warm-up results are discarded anyway: every rank
        restores its weights and optimiser state), the reason is written to stderr at once, not only in the warning behind the
        agreement.  Not covered: the 8 KB statistics all-reduces of sync_reward_bn inside a warm-up loss."""
        import sys
        print(f"[trainer] rank {fdist.rank()}: capture of the {kind} sub-update failed: {exc!r}", file=sys.stderr, flush=True)
        try:
            opt = self.policy_optimizer if which == "policy" else self.value_optimizer
            expected = 2 + (1 if os.environ.get("FLEX_GRAPH_AUDIT") == "1" else 0)
            missing = expected - (fdist.STATS["allreduce_calls"] - ar0)
            if missing > 0:
                if self.device.type == "cuda":
                    th.cuda.synchronize()
                n = sum(p.numel() for p in opt.param_groups[0]["params"])
                flat = th.zeros(n, dtype=th.float32, device=self.device)
                for _ in range(missing):
                    fdist.dist.all_reduce(flat, op=fdist.dist.ReduceOp.SUM)
            if self.allreduce_in_graph and fdist.STATS["agreements"] == ag0:
                fdist.all_agree(False, self.device)          # the peers' "is the one-graph form in on every rank?"
                if not self.sync_reward_bn:
                    self.allreduce_in_graph = False           # (they move to the split form on hearing this)
        except Exception as exc2:                             # (the group itself is broken: nothing left to align)
            print(f"[trainer] rank {fdist.rank()}: could not realign after the failed capture: {exc2!r}", file=sys.stderr, flush=True)

    def _capture_sub_update_body(self, kind, which, bs):
        from .replay_buffer import Transition
        buf = self.replay_buffer
        fields, plan = self._static_batch(kind, bs)
        # the value loss's reward-statistics pass rides in the launch that refreshes this batch (nets.offer_td_stats): offered
        # here, kept if the warm-up's loss takes it.  Not with the refresh on a side stream (the rider writes the statistics
        # the previous sub-update may still be reading) and not with cross-rank statistics (they are all-reduced per loss).
        td = offer = None
        from . import nets
        if (which == "value" and self.world == 1 and not self.pipeline_updates and self.device.type == "cuda"
                and os.environ.get("FLEX_TD_STATS_RIDER", "1") != "0" and th.is_tensor(fields.get("reward"))):
            blk = next((p[5] for p in plan if th.is_tensor(p[5]) and p[5].data_ptr() == fields["reward"].data_ptr()), None)
            if blk is not None:
                offer, targs = nets.offer_td_stats(blk)
                td = (blk, targs)
        try:
            g = self._capture_sub_update_graph(kind, which, bs, fields, plan, td)
        except Exception:
            if td is not None:
                nets.withdraw_td_stats(td[0])
            raise
        if td is not None and not offer["taken"]:
            nets.withdraw_td_stats(td[0])
            g["td"] = None
        return g

    def _capture_sub_update_graph(self, kind, which, bs, fields, plan, td):
        from .replay_buffer import Transition
        buf = self.replay_buffer
        buf.gather(plan, buf.warmup_slot(bs + buf.n_envs), td=td)  # real transitions for the warm-up steps: a window whose
        #                          next_state rows (N slots further on) exist too; no draw from the NumPy stream
        batch = Transition(**fields)
        out = {}
        # the warm-up steps are real optimiser steps: everything they touch is put back afterwards, IN PLACE (the graph
        # has the addresses baked in), so that capturing does not add updates to the schedule of model.py:43-50
        opt = self.policy_optimizer if which == "policy" else self.value_optimizer
        net_snap = {k: v.clone() for k, v in self.behaviour_net.state_dict().items()}
        had_state = {p: {k: (v.clone() if th.is_tensor(v) else v) for k, v in opt.state[p].items()}
                     for p in opt.param_groups[0]["params"] if p in opt.state}
        flat = None
        if self.world > 1:
            flat = th.zeros(sum(p.numel() for p in opt.param_groups[0]["params"]), dtype=th.float32, device=self.device)
            # (checked BEFORE the warm-up's real optimiser steps — ADVICE r03: raised behind them it left two unrestored
            #  critic updates on this rank only)
            if self.sync_reward_bn and not self.allreduce_in_graph:
                raise RuntimeError("cross-rank reward statistics need the all-reduces inside the update graph (nccl)")

        def restore():
            with th.no_grad():
                for k, v in self.behaviour_net.state_dict().items():
                    v.copy_(net_snap[k])
                for p in opt.param_groups[0]["params"]:
                    for k, v in opt.state.get(p, {}).items():
                        if th.is_tensor(v):
                            old = had_state.get(p, {}).get(k)
                            v.copy_(old) if old is not None else v.zero_()

        # Whatever happens between here and the end of the capture — a refused capture, an exception out of a loss — the
        # warm-up's steps (weights, RMSprop state, reward-BatchNorm running statistics) are undone: the invariant "capturing
        # adds no update to the schedule of model.py:43-50" also holds on the failure paths (eager fallback, value_cached
        # fallback), and with more than one rank a failure on one rank cannot leave the replicas apart.
        try:
            if os.environ.get("FLEX_GRAPH_AUDIT") == "1":     # the body about to be captured launches no ATen multi-block reduction
                from .util import audit_graph_body
                self.graph_audit = getattr(self, "graph_audit", {})
                self.graph_audit[kind] = audit_graph_body(lambda: self._sub_update(which, {}, batch, fresh_leaves=True, flat=flat))
            side = th.cuda.Stream()
            side.wait_stream(th.cuda.current_stream())
            with th.cuda.stream(side):
                for _ in range(2):                        # warm-up off the capturing stream (allocator, rocBLAS handles)
                    self._sub_update(which, out, batch, fresh_leaves=True, flat=flat)
            th.cuda.current_stream().wait_stream(side)
            if any(p[0] == "stack_ring" for p in plan):
                # `state` is a NaN placeholder read in place by ring-aware kernels only (nets.RING_VIEWS): a warm-up loss that is
                # not finite means some other consumer touched it — refuse the capture rather than train on NaN
                for v in out.values():
                    if th.is_tensor(v) and v.numel() == 1 and not bool(th.isfinite(v).all()):
                        raise RuntimeError("a consumer that cannot read the stacked-observation ring in place touched the placeholder")
            if self.world > 1:
                th.cuda.synchronize()                     # no collective of the warm-up is outstanding when capture begins
            graph = th.cuda.CUDAGraph()
            apply_graph = None
            out = {}
            if flat is None:
                with graph_capture(graph):
                    self._sub_update(which, out, batch, fresh_leaves=True)
            else:
                fused = False
                if self.allreduce_in_graph:
                    why = None
                    try:
                        with graph_capture(graph):
                            self._loss_and_grads(which, out, batch, fresh_leaves=True, flat=flat)
                            fdist.allreduce_flat(flat)
                            self._apply_grads(which, out, flat=flat)
                        fused = True
                    except Exception as exc:
                        why = exc
                    # one graph with the all-reduce inside it on EVERY rank, or the split form on every rank (VERDICT r04
                    # item 4: this used to be decided per rank)
                    th.cuda.synchronize()
                    if not fdist.all_agree(fused, self.device):
                        if self.sync_reward_bn:      # the statistics' all-reduce sits inside graph A: no split form — eager sub-updates
                            raise RuntimeError(f"all-reduces inside the sub-update graph could not be captured on every rank ({why})")
                        import warnings
                        warnings.warn(f"all-reduce inside the sub-update graph could not be captured on every rank ({why}); "
                                      "splitting the graph at it")
                        self.allreduce_in_graph = False
                        fused = False
                        graph = th.cuda.CUDAGraph()
                        out = {}
                if not fused:
                    with graph_capture(graph):
                        self._loss_and_grads(which, out, batch, fresh_leaves=True, flat=flat)
                    apply_graph = th.cuda.CUDAGraph()
                    with graph_capture(apply_graph, pool=graph.pool()):
                        self._apply_grads(which, out, flat=flat)
        finally:
            if self.device.type == "cuda":
                th.cuda.synchronize()                     # (a failed capture may leave warm-up work in flight)
            restore()
        # `batch` stays referenced: its constant fields (action_avail, ...) were allocated eagerly and are baked into the
        # graph by address; released, the allocator would hand their memory to the next eager tensor
        return dict(graph=graph, apply=apply_graph, flat=flat, plan=plan, stat=out, bs=bs, buf=buf, batch=batch, td=td,
                    allreduce_in_graph=bool(flat is not None and apply_graph is None))

    # kept for callers that hand over a batch themselves (trainer.py:81,99)
    def policy_transition_process(self, stat, trans):
        self._sub_update("policy", stat, trans)

    def value_transition_process(self, stat, trans):
        self._sub_update("value", stat, trans)

    # ---- one gradient step -----------------------------------------------------------------------
    def _sub_update(self, which, stat, batch, fresh_leaves=False, flat=None):
        """zero_grad -> loss -> backward -> (all-reduce) -> clip_grad_norm_(1.0) -> RMSprop (trainer.py:81-108).
        The policy loss carries the entropy bonus of trainer.py:47-57, a constant under the fixed std.
        ``flat``: a static bucket the gradients are gathered in (multi-rank graphed updates); None = per-call bucket."""
        self._loss_and_grads(which, stat, batch, fresh_leaves=fresh_leaves, flat=flat)
        if self.world > 1:
            if flat is not None:
                fdist.allreduce_flat(flat)
            else:
                opt = self.policy_optimizer if which == "policy" else self.value_optimizer
                fdist.allreduce_grads(opt.param_groups[0]["params"])
        self._apply_grads(which, stat, flat=flat)

    def _loss_and_grads(self, which, stat, batch, fresh_leaves=False, flat=None):
        """The loss this sub-update steps on and the gradients of ITS optimiser's parameters (left in ``p.grad``; with
        ``flat`` they are views of that bucket, in parameter order).

        ``fresh_leaves`` (graph capture): the loss is formed on detached aliases of the parameters.  Autograd keeps one
        gradient accumulator per parameter, bound to the stream it was first used on; if the caller still holds a
        graph built eagerly on the default stream (an evaluated policy output, say), backward would synchronise the
        capturing stream with the default stream — illegal during capture, and fatal in the HIP runtime.  Aliases are
        new leaves with accumulators of their own; their gradients ARE the parameters' gradients."""
        opt = self.policy_optimizer if which == "policy" else self.value_optimizer
        params = opt.param_groups[0]["params"]
        leaves = params
        if fresh_leaves:
            from torch.nn.utils.stateless import _reparametrize_module
            net = self.behaviour_net
            own = {id(p) for p in params}
            alias = {name: p.detach().requires_grad_(id(p) in own) for name, p in net.named_parameters()}
            by_param = {id(p): alias[name] for name, p in net.named_parameters()}
            leaves = [by_param[id(p)] for p in params]
            with _reparametrize_module(net, alias):
                policy_loss, value_loss, dist_params = self.get_loss(batch, need=which)
        else:
            policy_loss, value_loss, dist_params = self.get_loss(batch, need=which)
        opt.zero_grad()
        if which == "policy":
            loss = policy_loss
            if self.entr > 0:
                means, log_stds = dist_params
                # fixed policy std (model.py:121-123): the entropy bonus of trainer.py:47-57 is a constant that
                # Model.policy hands over with its log-std view — no elementwise kernels over [batch, n, act] per sub-update
                entropy = getattr(log_stds, "_flex_entropy", None)
                if entropy is None:
                    entropy = normal_entropy(means, log_stds.exp())
                    loss = loss - self.entr * entropy
                else:                                      # the constant's product with entr, formed once per device
                    key = (entropy.device, float(self.entr))
                    if key not in self._entr_terms:
                        self._entr_terms[key] = (self.entr * entropy).detach()
                    loss = loss - self._entr_terms[key]
                stat["mean_train_entropy"] = entropy.detach()
        else:
            loss = value_loss
        # gradients of THIS optimiser's parameters only: a plain backward() would also fill the other network's
        # .grad (the critic's first-layer weight gradient is the largest GEMM of a policy step) just to have it
        # zeroed by that optimiser's next zero_grad (trainer.py:82,100)
        seed = None
        if loss.is_cuda and loss.dim() == 0 and loss.dtype == th.float32:
            from .util import unit_seed
            seed = unit_seed(loss.device)             # recognised by the loss nodes: no ones-fill, no multiply by one
        grads = th.autograd.grad(loss, leaves, grad_outputs=seed, allow_unused=True)
        if flat is None:
            for p, g in zip(params, grads):
                p.grad = g
        else:
            views, off = [], 0
            for p in params:
                views.append(flat[off:off + p.numel()].view_as(p))
                off += p.numel()
            th._foreach_copy_(views, [g if g is not None else th.zeros_like(v) for g, v in zip(grads, views)])
            for p, v in zip(params, views):
                p.grad = v
        stat[f"mean_train_{which}_loss"] = loss.detach()

    def _apply_grads(self, which, stat, flat=None):
        """clip_grad_norm_ + RMSprop step (trainer.py:86-90,103-107), after the all-reduce; one HIP launch on the GPU."""
        opt = self.policy_optimizer if which == "policy" else self.value_optimizer
        params = opt.param_groups[0]["params"]
        if flat is not None and self.world > 1:
            flat.mul_(1.0 / self.world)               # the bucket holds the SUM over ranks
        refresh = getattr(self, "_next_refresh", None)
        self._next_refresh = None                     # (taken: _ensure_event_graph launches it itself otherwise)
        grad_norm = clip_and_step(opt, params, self.args.grad_clip_eps, refresh=refresh)
        stat[f"mean_train_{which}_grad_norm"] = grad_norm.detach()

    # ---- episode loop hooks (train_agent.py:125-146) --------------------------------------------
    def run(self, stat, episode):
        net = self.behaviour_net
        net.train_process(stat, self)
        if episode == 0 or episode % self.args.eval_freq == self.args.eval_freq - 1:      # trainer.py:122-124
            net.evaluation(stat, self)
        for key, val in list(stat.items()):
            if isinstance(val, th.Tensor):
                stat[key] = float(val.item())

    def logging(self, stat):
        if self.logger is not None:
            for key, val in stat.items():
                self.logger.add_scalar("data/" + key, val, self.episodes)

    def print_info(self, stat):
        lines = [f"\nEpisode: {self.episodes}"] + [f"{key}: {float(val):2.4f}" for key, val in stat.items()]
        train_logger.info("\n".join(lines))
