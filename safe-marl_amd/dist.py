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
        dist.broadcast(flat, src=src)
        off = 0
        for t in tensors:
            n = t.numel()
            t.copy_(flat[off:off + n].view_as(t))
            off += n


def allreduce_grads(params):
    """Mean of the gradients over ranks through ONE flattened bucket (sum, then scale)."""
    grads = [p.grad for p in params if p.grad is not None]
    if not grads:
        return
    flat = th.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat.div_(dist.get_world_size())
    off = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[off:off + n].view_as(g))
        off += n


def allreduce_flat(flat):
    """SUM over ranks of an already-flat gradient bucket, in place (the caller scales by 1/world — inside its HIP graph
    when the update is graphed).  One ncclAllReduce on RCCL's stream, ordered after the current stream."""
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)


def shard_envs(total_envs):
    """Contiguous block of environments for this rank."""
    w, r = world_size(), rank()
    per = total_envs // w
    return r * per, per
