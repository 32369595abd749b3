#!/usr/bin/env python3
"""Two (or more) data-parallel ranks of the TRAINING loop on ONE GPU, gloo instead of RCCL for the exchange (RCCL
refuses two ranks on one device): rehearses BASELINE config 5's control flow — per-rank env shard and replay shard,
broadcast replicas, and per sub-update graph A (losses + backward into the flat bucket) -> all-reduce -> graph B
(1/world, clip, RMSprop) — on the one-GPU box.  Launched by tests/test_dist_gpu.py as

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P \
        tools/dist_rehearsal.py --out DIR --graph-updates 1

Every rank writes DIR/rank<r>.npz: final behaviour / target weights, optimiser state, statistics."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "examples"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--graph-updates", type=int, default=1)
    ap.add_argument("--envs", type=int, default=256)
    ap.add_argument("--force-cached-rank", type=int, default=-1,
                    help="a SECOND episode in which rank R's update events take the filed-bootstrap form and the other ranks' do not "
                         "(trainer.force_bootstrap_choice), after the replay's stacked ring has been re-allocated — every graph "
                         "captured against the old ring is stale and is recaptured (warm-up all-reduces, agreement): all ranks "
                         "must do that at the same event whatever form they go on to choose")
    ap.add_argument("--alg", default="maddpg", choices=["maddpg", "safemaddpg"])
    ap.add_argument("--force-from-start", type=int, default=0,
                    help="with --force-cached-rank: the forced choice holds from the FIRST update event (no second episode, no ring "
                         "re-allocation by hand) — SAFEMADDPG's plain value graph goes stale when the cached form's capture creates the "
                         "stacked ring, inside that very event")
    ap.add_argument("--fail-capture-rank", type=int, default=-1,
                    help="rank R's FIRST sub-update capture raises before it has done anything (injected here, not in the product): "
                         "R has to catch up with its peers' warm-up all-reduces and every rank falls back to eager sub-updates")
    a = ap.parse_args()
    import torch
    import torch.distributed as dist
    import safe_marl_amd  # noqa: F401
    from train_maddpg import DEFAULT_ALG_ARGS
    from safe_marl_amd.flex_env import VecFlexProvisionEnv
    from safe_marl_amd.learner import MADDPG, SAFEMADDPG
    from safe_marl_amd.network import create_network
    from safe_marl_amd.series import make_synthetic_series
    from safe_marl_amd.trainer import PGTrainer
    from safe_marl_amd.util import convert

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(0)
    if world > 1:
        dist.init_process_group("gloo")
    net = create_network()
    series = make_synthetic_series(net, n_days=30)
    alg = dict(DEFAULT_ALG_ARGS)
    alg.update(alg=a.alg, v_min=0.9, v_max=1.1, agent_num=5, obs_size=144, state_size=110, action_dim=4, behaviour_update_freq=30,
               target_update_freq=60)
    torch.manual_seed(50 + rank)                     # different initial weights per rank: the broadcast must fix that
    np.random.seed(70 + rank)                        # a different replay window per rank
    env = VecFlexProvisionEnv({"alg": "safemaddpg"} if a.alg == "safemaddpg" else {}, a.envs, device="cuda:0", net=net, series=series, seed=1234 + 1000 * rank, warm_start=True)
    tr = PGTrainer(convert(alg), {"maddpg": MADDPG, "safemaddpg": SAFEMADDPG}[a.alg], env, None, replay_capacity=a.envs * 96 * 2, graph_updates=bool(a.graph_updates))
    if rank == a.fail_capture_rank:
        body, hit = tr._capture_sub_update_body, []

        def failing_body(*args, **kw):
            if not hit:
                hit.append(1)
                raise RuntimeError("injected capture failure (tools/dist_rehearsal.py --fail-capture-rank)")
            return body(*args, **kw)
        tr._capture_sub_update_body = failing_body
    if a.force_cached_rank >= 0 and a.force_from_start:
        tr.force_bootstrap_choice = rank == a.force_cached_rank
    stat = {}
    tr.behaviour_net.train_process(stat, tr)         # 95 vector steps: update events at 30, 60, 90 (33 sub-updates)
    if a.force_cached_rank >= 0 and not a.force_from_start:
        buf = tr.replay_buffer
        gen0 = getattr(buf, "stack_gen", None)
        if getattr(buf, "stack_ring", None) is not None:
            buf.enable_stacked_ring(buf.stack_tail + buf.n_envs)       # a longer tail: a NEW ring, the captured graphs go stale
        assert getattr(buf, "stack_gen", None) != gen0 or gen0 is None
        tr.force_bootstrap_choice = rank == a.force_cached_rank
        tr.behaviour_net.train_process(stat, tr)
    torch.cuda.synchronize()
    net_ = tr.behaviour_net
    out = {"w": torch.cat([p.detach().reshape(-1) for p in net_.parameters()]).cpu().numpy(),
           "sq": torch.cat([s["square_avg"].reshape(-1) for o in (tr.value_optimizer, tr.policy_optimizer)
                            for s in o.state.values()]).cpu().numpy(),
           "vloss": np.float64(float(stat["mean_train_value_loss"])), "reward": np.float64(stat["mean_train_reward"]),
           "graphs": np.array(sorted(tr._update_graphs)), "split": np.array([g["apply"] is not None for g in tr._update_graphs.values()]),
           "steps": np.int64(tr.steps), "cached_events": np.int64(getattr(tr, "bootstrap_cached_events", 0)),
           "ring_gen": np.int64(getattr(tr.replay_buffer, "stack_gen", 0) or 0)}
    np.savez(os.path.join(a.out, f"rank{rank}.npz"), **out)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
