"""Multi-rank path on CPU (gloo, world_size 2): the gradient bucket all-reduce and replica broadcast the
8-GPU run uses over RCCL."""
import json
import os
import socket

import numpy as np
import pytest
import torch as th
import torch.distributed as dist
import torch.multiprocessing as mp

G = os.path.join(os.path.dirname(__file__), "golden")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    import safe_marl_amd  # noqa: F401
    from safe_marl_amd import dist as fdist
    from safe_marl_amd.learner import MADDPG
    from safe_marl_amd.replay_buffer import Transition
    from safe_marl_amd.trainer import PGTrainer
    from safe_marl_amd.util import convert
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    args = convert(json.load(open(os.path.join(G, "learner_args.json"))))

    class Env:
        n_envs = 1

    th.manual_seed(100 + rank)                    # different initial weights per rank on purpose
    trainer = PGTrainer(args, MADDPG, Env(), None)
    w0 = th.cat([p.detach().reshape(-1) for p in trainer.behaviour_net.parameters()])
    z = np.load(os.path.join(G, "learner_batch.npz"))
    lo, hi = (0, 16) if rank == 0 else (16, 32)   # each rank sees its own half of the batch
    batch = Transition(**{k: th.from_numpy(z[k]).float()[lo:hi] for k in Transition._fields})
    stat = {}
    # reward BatchNorm uses per-rank batch statistics (documented deviation): switch it off here so
    # that the two half-batches average to the full-batch gradient exactly
    trainer.behaviour_net.args = trainer.args = args._replace(reward_normalisation=False)
    trainer.value_transition_process(stat, batch)
    trainer.policy_transition_process(stat, batch)
    w1 = th.cat([p.detach().reshape(-1) for p in trainer.behaviour_net.parameters()])
    start, per = fdist.shard_envs(8192)
    out[rank] = dict(w0=w0.numpy(), w1=w1.numpy(), vnorm=float(stat["mean_train_value_grad_norm"]), shard=(start, per))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_stay_identical_and_average_gradients():
    import safe_marl_amd  # noqa: F401
    from safe_marl_amd.learner import MADDPG
    from safe_marl_amd.replay_buffer import Transition
    from safe_marl_amd.trainer import PGTrainer
    from safe_marl_amd.util import convert
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    a, b = out[0], out[1]
    assert np.array_equal(a["w0"], b["w0"])        # rank 0's weights were broadcast at construction
    assert np.array_equal(a["w1"], b["w1"])        # and the replicas stay bit-identical after an optimiser step
    assert a["shard"] == (0, 4096) and b["shard"] == (4096, 4096)
    # single-process reference: the same step on the full 32-sample batch from the same initial weights
    args = convert(json.load(open(os.path.join(G, "learner_args.json"))))._replace(reward_normalisation=False)

    class Env:
        n_envs = 1

    trainer = PGTrainer(args, MADDPG, Env(), None)
    with th.no_grad():
        off = 0
        for p in trainer.behaviour_net.parameters():
            p.copy_(th.from_numpy(a["w0"][off:off + p.numel()]).view_as(p))
            off += p.numel()
    z = np.load(os.path.join(G, "learner_batch.npz"))
    batch = Transition(**{k: th.from_numpy(z[k]).float() for k in Transition._fields})
    stat = {}
    trainer.value_transition_process(stat, batch)
    trainer.policy_transition_process(stat, batch)
    w = th.cat([p.detach().reshape(-1) for p in trainer.behaviour_net.parameters()]).numpy()
    assert np.allclose(w, a["w1"], atol=2e-6)
    assert abs(float(stat["mean_train_value_grad_norm"]) - a["vnorm"]) < 1e-4 * max(1.0, a["vnorm"])
