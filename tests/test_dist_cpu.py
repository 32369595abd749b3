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


def _worker(rank, world, port, out, sync_bn=False):
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
    trainer = PGTrainer(args, MADDPG, Env(), None, sync_reward_bn=sync_bn)
    w0 = th.cat([p.detach().reshape(-1) for p in trainer.behaviour_net.parameters()])
    z = np.load(os.path.join(G, "learner_batch.npz"))
    lo, hi = (0, 16) if rank == 0 else (16, 32)   # each rank sees its own half of the batch
    batch = Transition(**{k: th.from_numpy(z[k]).float()[lo:hi] for k in Transition._fields})
    stat = {}
    # reward BatchNorm uses per-rank batch statistics by default (documented deviation): switched off here so
    # that the two half-batches average to the full-batch gradient exactly — or left ON with the cross-rank statistics
    # (sync_reward_bn), under which the same must hold
    if not sync_bn:
        trainer.behaviour_net.args = trainer.args = args._replace(reward_normalisation=False)
    trainer.value_transition_process(stat, batch)
    trainer.policy_transition_process(stat, batch)
    w1 = th.cat([p.detach().reshape(-1) for p in trainer.behaviour_net.parameters()])
    start, per = fdist.shard_envs(8192)
    bn = trainer.behaviour_net.batchnorm
    out[rank] = dict(w0=w0.numpy(), w1=w1.numpy(), vnorm=float(stat["mean_train_value_grad_norm"]), shard=(start, per),
                     bn_mean=bn.running_mean.numpy().copy(), bn_var=bn.running_var.numpy().copy(),
                     bn_n=int(bn.num_batches_tracked))
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


def test_cross_rank_reward_statistics_make_two_half_batches_the_full_batch():
    """SURVEY.md 8e / VERDICT r02 item 8b: with ``sync_reward_bn`` the reward BatchNorm's batch statistics are summed over
    the ranks (2 x 5 moments per normalisation), so two ranks on half a batch each take the step one process takes on the
    whole batch WITH the normalisation on — weights, gradient norm and the module's running statistics."""
    import safe_marl_amd  # noqa: F401
    from safe_marl_amd.learner import MADDPG
    from safe_marl_amd.replay_buffer import Transition
    from safe_marl_amd.trainer import PGTrainer
    from safe_marl_amd.util import convert
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, port, out, True), nprocs=2, join=True)
    a, b = out[0], out[1]
    assert np.array_equal(a["w1"], b["w1"]) and np.array_equal(a["bn_mean"], b["bn_mean"])
    args = convert(json.load(open(os.path.join(G, "learner_args.json"))))
    assert args.reward_normalisation

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
    bn = trainer.behaviour_net.batchnorm
    assert int(bn.num_batches_tracked) == a["bn_n"] == 2          # one normalisation per get_loss call (value, policy)
    assert np.allclose(bn.running_mean.numpy(), a["bn_mean"], rtol=1e-5, atol=1e-7)
    assert np.allclose(bn.running_var.numpy(), a["bn_var"], rtol=1e-5, atol=1e-7)
    # ... and per-rank statistics (the default) do NOT give the full-batch step: the deviation DESIGN.md documents
    out2 = mgr.dict()
    mp.spawn(_worker_per_rank_bn, args=(2, _free_port(), out2), nprocs=2, join=True)
    assert not np.allclose(out2[0]["w1"], a["w1"], atol=2e-6)


def _worker_per_rank_bn(rank, world, port, out):
    """As _worker with the normalisation ON and per-rank statistics (sync off)."""
    import safe_marl_amd  # noqa: F401
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

    th.manual_seed(100 + rank)
    trainer = PGTrainer(args, MADDPG, Env(), None, sync_reward_bn=False)
    z = np.load(os.path.join(G, "learner_batch.npz"))
    lo, hi = (0, 16) if rank == 0 else (16, 32)
    batch = Transition(**{k: th.from_numpy(z[k]).float()[lo:hi] for k in Transition._fields})
    stat = {}
    trainer.value_transition_process(stat, batch)
    trainer.policy_transition_process(stat, batch)
    out[rank] = dict(w1=th.cat([p.detach().reshape(-1) for p in trainer.behaviour_net.parameters()]).numpy())
    dist.barrier()
    dist.destroy_process_group()


class _StubVecEnv:
    """The slice of VecFlexProvisionEnv that Model._train_process_vec touches, on CPU tensors: independent synthetic
    environments (per-rank seed), so that the two ranks see DIFFERENT data and only the gradient all-reduce can keep the
    replicas identical.  The physics is irrelevant here (the HIP env has no CPU path); the control flow is the product's."""
    handle = True

    def __init__(self, n_envs, seed, n_agents=5, obs_size=144, episode_limit=25):
        self.n_envs, self.n_agents, self.obs_size, self.episode_limit = n_envs, n_agents, obs_size, episode_limit
        self.gen = th.Generator().manual_seed(seed)
        self.t = 0
        self.obs = th.zeros(n_envs, n_agents, obs_size)
        self.reward = th.zeros(n_envs, dtype=th.float64)
        self.done = th.zeros(n_envs, dtype=th.uint8)
        self.failed = th.zeros(n_envs, dtype=th.uint8)
        self.info = th.zeros(n_envs, 8, dtype=th.float64)

    def _draw_obs(self):
        self.obs = 0.3 * th.randn(self.n_envs, self.n_agents, self.obs_size, generator=self.gen)

    def reset(self):
        self.t = 0
        self._draw_obs()
        return self.obs

    def safety_project(self, proposed, s_p, s_q, beta, v_min, v_max, penalty=1000.0, env_action_range=None, want_hit=True):
        """A CPU stand-in with the HIP layer's interface and output layout (flex_env.safety_project: [N, 4 n] float64,
        type-major, safemaddpg.py:297): percentage / charge / discharge clipped at zero, q shifted by the predictor's
        offset — enough for SAFEMADDPG's control flow (safemaddpg.py:90-111) to run through the data-parallel trainer."""
        p = proposed.detach().double().reshape(self.n_envs, self.n_agents, 4)
        adj = th.cat([p[:, :, 0].clamp_min(0), p[:, :, 1].clamp_min(0), p[:, :, 2].clamp_min(0),
                      p[:, :, 3] + th.as_tensor(beta, dtype=th.float64) * 0.0], dim=1)
        return adj, th.zeros(self.n_envs, dtype=th.uint8)

    def step(self, actions, fuse_obs=True, auto_reset=True):
        self.t += 1
        self.reward = (0.03 + 0.1 * actions.double().mean((1, 2)) + 0.02 * th.randn(self.n_envs, generator=self.gen).double())
        self.done = th.full((self.n_envs,), int(self.t >= self.episode_limit - 1), dtype=th.uint8)
        self.info = self.reward.unsqueeze(1).expand(-1, 8).contiguous()
        self._draw_obs()
        return self.reward, self.done, self.info


def _train_worker(rank, world, port, out, alg="maddpg", sync_bn=False):
    import safe_marl_amd  # noqa: F401
    from safe_marl_amd.learner import MADDPG, SAFEMADDPG
    from safe_marl_amd.trainer import PGTrainer
    from safe_marl_amd.util import convert
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    d = json.load(open(os.path.join(G, "learner_args.json")))
    d.update(behaviour_update_freq=10, target_update_freq=20, alg=alg, v_min=0.9, v_max=1.1)
    args = convert(d)
    th.manual_seed(100 + rank)                    # different initial weights per rank: the constructor broadcast fixes that
    np.random.seed(7 + rank)                      # ... and a different replay window per rank (per-rank shard)
    env = _StubVecEnv(8, seed=1000 + rank)
    if alg == "safemaddpg":                       # trainer.py:16-28 hands SAFEMADDPG the env; a fixed predictor (no fit on the CPU)
        import functools
        pred = (th.full((5,), -0.05, dtype=th.float64), th.full((5,), -0.03, dtype=th.float64), th.full((5,), 1.0, dtype=th.float64))
        cls = functools.partial(SAFEMADDPG, predictor=pred)
        cls.__name__ = "SAFEMADDPG"
    else:
        cls = MADDPG
    trainer = PGTrainer(args, cls, env, None, batch_scale=2, sync_reward_bn=sync_bn)
    w0 = th.cat([p.detach().reshape(-1) for p in trainer.behaviour_net.parameters()]).clone()
    stat = {}
    trainer.behaviour_net.train_process(stat, trainer)        # 24 vector steps: update events at steps 10 and 20
    w1 = th.cat([p.detach().reshape(-1) for p in trainer.behaviour_net.parameters()])
    tgt = th.cat([p.detach().reshape(-1) for p in trainer.behaviour_net.target_net.parameters()])
    bn = trainer.behaviour_net.batchnorm
    from safe_marl_amd import dist as fdist
    div = fdist.replica_divergence(trainer.behaviour_net) if world > 1 else {"params": 0.0, "buffers": 0.0}
    out[rank] = dict(w0=w0.numpy(), w1=w1.numpy(), tgt=tgt.numpy(), steps=trainer.steps, buf=len(trainer.replay_buffer.buffer),
                     allreduce_calls=fdist.STATS["allreduce_calls"], allreduce_bytes=fdist.STATS["allreduce_bytes"], div=div,
                     vloss=float(stat["mean_train_value_loss"]), reward=float(stat["mean_train_reward"]),
                     bn_mean=bn.running_mean.numpy().copy(), bn_var=bn.running_var.numpy().copy(), bn_n=int(bn.num_batches_tracked),
                     model=type(trainer.behaviour_net).__name__)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_train_process_on_two_ranks_keeps_replicas_identical():
    """model.py:198-267 + model.py:40-71 under data parallelism (BASELINE config 5's control flow on gloo): every rank
    rolls out its own env shard into its own replay shard, the 22 sub-updates of two update events all-reduce their
    gradient bucket before the clip, and behaviour AND target replicas end bit-identical although the data differ."""
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_train_worker, args=(2, port, out), nprocs=2, join=True)
    a, b = out[0], out[1]
    assert a["steps"] == b["steps"] == 24 and a["buf"] == b["buf"] == 24 * 8
    assert np.array_equal(a["w0"], b["w0"])
    assert not np.array_equal(a["w0"], a["w1"])                    # updates happened
    assert np.array_equal(a["w1"], b["w1"]) and np.array_equal(a["tgt"], b["tgt"])
    assert a["vloss"] != b["vloss"] and a["reward"] != b["reward"]          # the ranks really saw different data
    # a lone rank 0 from the same start, without the exchange, ends somewhere else: the all-reduce mattered
    solo = mgr.dict()
    mp.spawn(_train_worker, args=(1, _free_port(), solo), nprocs=1, join=True)
    assert np.array_equal(solo[0]["w0"], a["w0"]) and not np.array_equal(solo[0]["w1"], a["w1"])


@pytest.mark.parametrize("alg,sync_bn", [("safemaddpg", False), ("maddpg", True), ("safemaddpg", True)])
def test_train_process_on_two_ranks_safemaddpg_and_cross_rank_reward_statistics(alg, sync_bn):
    """the two-rank train_process test for SAFEMADDPG (BASELINE config 4's algorithm under config 5's data
    parallelism: the safety layer between policy and env, safemaddpg.py:90-111, through a CPU stand-in with the HIP layer's
    interface) and with ``sync_reward_bn`` — every reward normalisation of the 22 sub-updates sums its moments over the ranks
    first: replicas AND the reward BatchNorm's running statistics end bit-identical on ranks that saw different data; with
    per-rank statistics (the default) the running statistics differ."""
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_train_worker, args=(2, port, out, alg, sync_bn), nprocs=2, join=True)
    a, b = out[0], out[1]
    assert a["model"] == b["model"] == ("SAFEMADDPG" if alg == "safemaddpg" else "MADDPG")
    assert a["steps"] == b["steps"] == 24 and a["buf"] == b["buf"] == 24 * 8
    assert np.array_equal(a["w0"], b["w0"]) and not np.array_equal(a["w0"], a["w1"])
    assert np.array_equal(a["w1"], b["w1"]) and np.array_equal(a["tgt"], b["tgt"])
    assert a["vloss"] != b["vloss"] and a["reward"] != b["reward"]          # the ranks really saw different data
    assert a["bn_n"] == b["bn_n"] == 22                                     # one normalisation per sub-update
    if sync_bn:
        assert np.array_equal(a["bn_mean"], b["bn_mean"]) and np.array_equal(a["bn_var"], b["bn_var"])
    else:
        assert not np.array_equal(a["bn_mean"], b["bn_mean"])               # per-rank statistics: the documented deviation


def test_train_process_on_four_ranks_counts_its_all_reduces_and_keeps_replicas_identical():
    """world 4, and the figures bench.py prints for an N > 1 leg checked where they can be checked — every
    rank counts exactly one all-reduce of one flat bucket per gradient step (22 = two update events x (10 value + 1 policy));
    replica_divergence reports 0.0 for the parameters on every rank and a non-zero figure for the per-rank reward statistics."""
    world = 4
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_train_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    r = [out[i] for i in range(world)]
    for x in r[1:]:
        assert np.array_equal(x["w1"], r[0]["w1"]) and np.array_equal(x["tgt"], r[0]["tgt"])
        assert x["vloss"] != r[0]["vloss"]
    n_value, n_policy = 52097, 34948                       # critic / actor parameters of the default config
    for x in r:
        assert x["allreduce_calls"] == 22
        assert x["allreduce_bytes"] == 4 * (20 * n_value + 2 * n_policy)
        assert x["div"]["params"] == 0.0 and x["div"]["buffers"] > 0.0


def _control_worker(rank, world, port, out):
    import safe_marl_amd  # noqa: F401
    from safe_marl_amd import dist as fdist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res = {"shard": fdist.shard_envs(world * 4096), "seed": 1234 + 1000 * fdist.rank(), "world": fdist.world_size(),
           "backend": fdist.backend()}
    # a decision every rank must take together: one rank's failure moves everybody to the fallback
    res["agree_all_ok"] = fdist.all_agree(True)
    res["agree_one_bad"] = fdist.all_agree(rank != 5)
    lin = th.nn.Linear(7, 3)
    th.manual_seed(rank)
    with th.no_grad():
        lin.weight.normal_()
    res["div_before"] = fdist.replica_divergence(lin)["params"]
    fdist.broadcast_module(lin)
    res["div_after"] = fdist.replica_divergence(lin)["params"]
    flat = th.full((10,), float(rank + 1))
    fdist.allreduce_flat(flat)
    res["sum"] = float(flat[0])
    res["stats"] = dict(fdist.STATS)
    out[rank] = res
    dist.barrier()
    dist.destroy_process_group()


def test_world_8_control_flow_on_gloo():
    """BASELINE config 5's host-side control flow at its real world size (the 8-GPU run itself is the driver's): contiguous
    env shards of 4096, per-rank seeds 1234 + 1000 rank (bench.py), the cross-rank agreement that decides the update-graph
    form, replica broadcast / divergence check, the flat-bucket all-reduce and its counters."""
    world = 8
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_control_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    for r in range(world):
        x = out[r]
        assert x["world"] == 8 and x["backend"] == "gloo"
        assert x["shard"] == (4096 * r, 4096) and x["seed"] == 1234 + 1000 * r
        assert x["agree_all_ok"] is True and x["agree_one_bad"] is False
        assert x["div_before"] > 0.0 and x["div_after"] == 0.0
        assert x["sum"] == sum(range(1, 9))
        assert x["stats"]["allreduce_calls"] == 1 and x["stats"]["allreduce_bytes"] == 40 and x["stats"]["agreements"] == 2
