"""Multi-GPU glue: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on the GPU
box, "gloo" in CPU tests).  Environments are independent, so the data path has no collective; the only
exchange is the gradient all-reduce of the policy (34 948 fp32) or value (52 097 fp32) parameters —
one flat bucket of <= 208 KB per optimiser step.  At that size the collective is latency-bound
(launch + sync, not the 153 GB/s xGMI links), so ONE bucket per step matters and bandwidth does not
(SURVEY.md §8e)."""
from __future__ import annotations

import torch as th
import torch.distributed as dist


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def backend():
    return str(dist.get_backend()) if dist.is_available() and dist.is_initialized() else ""


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def broadcast_module(module, src=0):
    """Identical replicas: copy rank ``src``'s parameters and buffers everywhere (one flat bucket per dtype)."""
    with th.no_grad():
        tensors = [t for t in list(module.parameters()) + list(module.buffers()) if t.is_floating_point()]
        if not tensors:
            return
        flat = th.cat([t.reshape(-1).float() for t in tensors])
        _log_collective("broadcast_module", flat.numel() * 4)
        dist.broadcast(flat, src=src)
        off = 0
        for t in tensors:
            n = t.numel()
            t.copy_(flat[off:off + n].view_as(t))
            off += n


# What this rank has put through the exchange step so far: eager calls are counted where they are made, an all-reduce
# captured inside an update graph once per REPLAY of that graph (trainer._replay).  bench.py reports them per leg so that
# the first multi-GPU run can be checked from its own output.
STATS = {"allreduce_calls": 0, "allreduce_bytes": 0, "agreements": 0}


_COLL_LOG = None


def _log_collective(what, n_bytes):
    """FLEX_COLL_LOG=<prefix>: every collective this module issues, one line per call with the caller's frames, appended to
    <prefix>.rank<r>.log — two ranks' files are compared line by line when their collectives stop matching."""
    global _COLL_LOG
    import os
    prefix = os.environ.get("FLEX_COLL_LOG")
    if not prefix:
        return
    import traceback
    if _COLL_LOG is None:
        _COLL_LOG = open(f"{prefix}.rank{rank()}.log", "a", buffering=1)
    frames = traceback.extract_stack(limit=7)[:-2]
    _COLL_LOG.write(f"{what} {n_bytes} B <- " + " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}:{f.name}" for f in reversed(frames)) + "\n")


def note_allreduce(n_bytes, calls=1):
    STATS["allreduce_calls"] += int(calls)
    STATS["allreduce_bytes"] += int(n_bytes) * int(calls)


def all_agree(ok, device=None):
    """True iff EVERY rank passes ``ok`` (an all-reduce(MIN) of one flag; trivially ``ok`` on one rank).  Decisions that
    change what a rank puts on the wire afterwards — one update graph with the all-reduce inside it, two graphs around an
    eager one, eager sub-updates — are taken through this, never locally: a capture that fails on one rank only must move
    every rank to the fallback, or their collectives stop matching."""
    if world_size() <= 1:
        return bool(ok)
    dev = device if (device is not None and backend() == "nccl") else th.device("cpu")
    flag = th.tensor([1 if ok else 0], dtype=th.int32, device=dev)
    _log_collective(f"all_agree({bool(ok)})", 4)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    STATS["agreements"] += 1
    return bool(int(flag.item()))


def replica_divergence(module, src=0):
    """max |theta - theta_src| over the PARAMETERS of ``module`` and, separately, over its floating-point buffers, maximised
    over the ranks.  Parameters must come out 0.0 exactly: every rank applies the same averaged gradient to the same
    weights.  Buffers hold the reward BatchNorm's running statistics, which are per-rank unless sync_reward_bn is set."""
    out = []
    with th.no_grad():
        for tensors in (list(module.parameters()), [b for b in module.buffers() if b.is_floating_point()]):
            if not tensors:
                out.append(0.0)
                continue
            mine = th.cat([t.detach().reshape(-1).float() for t in tensors])
            if world_size() <= 1:
                out.append(0.0)
                continue
            ref = mine.clone()
            _log_collective("replica_divergence broadcast + max", ref.numel() * 4)
            dist.broadcast(ref, src=src)
            d = (mine - ref).abs().max().reshape(1)
            dist.all_reduce(d, op=dist.ReduceOp.MAX)
            out.append(float(d.item()))
    return {"params": out[0], "buffers": out[1]}


def allreduce_grads(params):
    """Mean of the gradients over ranks through ONE flattened bucket (sum, then scale)."""
    grads = [p.grad for p in params if p.grad is not None]
    if not grads:
        return
    flat = th.cat([g.reshape(-1) for g in grads])
    _log_collective("allreduce_grads", flat.numel() * flat.element_size())
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    note_allreduce(flat.numel() * flat.element_size())
    flat.div_(dist.get_world_size())
    off = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[off:off + n].view_as(g))
        off += n


def allreduce_flat(flat):
    """SUM over ranks of an already-flat gradient bucket, in place (the caller scales by 1/world — inside its HIP graph
    when the update is graphed).  One ncclAllReduce on RCCL's stream, ordered after the current stream."""
    _log_collective("allreduce_flat", flat.numel() * flat.element_size())
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    if not th.cuda.is_available() or not th.cuda.is_current_stream_capturing():
        note_allreduce(flat.numel() * flat.element_size())          # (a captured one is counted per replay)


def shard_envs(total_envs):
    """Contiguous block of environments for this rank."""
    w, r = world_size(), rank()
    per = total_envs // w
    return r * per, per
