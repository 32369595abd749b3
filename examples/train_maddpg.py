#!/usr/bin/env python3
"""Vectorised (SAFE)MADDPG training on the HIP environment — BASELINE.json configs 3-5.

    python examples/train_maddpg.py --alg maddpg --envs 4096 --episodes 3
    python examples/train_maddpg.py --alg safemaddpg --envs 8192 --episodes 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 examples/train_maddpg.py ...

One "episode" is 95 vector steps.  Env shards are independent; with more than one rank the
only collective is the flat gradient bucket all-reduce per optimiser step (RCCL over xGMI).
Prints one JSON line with whole-job env-steps/s including policy inference, replay writes and the
11 gradient steps per 60 vector steps of model.py:40-71.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

DEFAULT_ALG_ARGS = dict(  # madrl/args/default.yaml merged with alg_args/maddpg.yaml
    gumbel_softmax=False, epsilon_softmax=False, softmax_eps=None, episodic=False, cuda=True, grad_clip_eps=1.0,
    save_model_freq=40, replay_warmup=0, policy_lrate=1.0e-4, value_lrate=1.0e-4, mixer_lrate=None, target=True,
    target_lr=0.1, entr=1.0e-3, max_steps=240, batch_size=32, replay=True, replay_buffer_size=5.0e3, agent_type="rnn",
    agent_id=True, shared_params=True, layernorm=True, mixer=False, gaussian_policy=False, LOG_STD_MIN=0.0,
    LOG_STD_MAX=0.5, fixed_policy_std=1.0, hid_activation="relu", init_type="normal", init_std=0.1,
    action_enforcebound=True, double_q=True, clip_c=1.0, gamma=0.99, hid_size=64, continuous=True,
    normalize_advantages=False, train_episodes_num=400, behaviour_update_freq=60, target_update_freq=120,
    policy_update_epochs=1, value_update_epochs=10, mixer_update_epochs=None, reward_normalisation=True, eval_freq=20,
    num_eval_episodes=10, action_low=0, action_high=1.0, action_bias=0.0, action_scale=1.0,
)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--alg", choices=["maddpg", "safemaddpg", "matd3", "iddpg"], default="maddpg")
    ap.add_argument("--envs", type=int, default=4096, help="envs per GPU")
    ap.add_argument("--episodes", type=int, default=3)
    ap.add_argument("--agents", type=int, default=5, choices=[3, 5])
    ap.add_argument("--batch-scale", type=int, default=None)
    a = ap.parse_args()

    import torch
    import safe_marl_amd  # noqa: F401
    from safe_marl_amd.flex_env import VecFlexProvisionEnv
    from safe_marl_amd.learner import IDDPG, MADDPG, MATD3, SAFEMADDPG
    from safe_marl_amd.network import create_network
    from safe_marl_amd.series import make_synthetic_series
    from safe_marl_amd.trainer import PGTrainer
    from safe_marl_amd.util import convert

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    blds = [5, 10, 15, 20, 25] if a.agents == 5 else [5, 15, 25]
    env_args = {"buildings": blds, "pv_nodes": blds, "ess_nodes": blds}
    if a.alg == "safemaddpg":
        env_args["alg"] = "safemaddpg"
    net = create_network(env_args)
    series = make_synthetic_series(net, n_days=365)
    env = VecFlexProvisionEnv(env_args, a.envs, device=f"cuda:{local}", net=net, series=series,
                              seed=1234 + 1000 * rank, warm_start=True)
    alg = dict(DEFAULT_ALG_ARGS)
    alg.update(alg=a.alg, agent_num=env.n_agents, obs_size=env.obs_size, state_size=env.state_size,
               action_dim=4, v_min=0.9, v_max=1.1)
    args = convert(alg)
    torch.manual_seed(0)
    trainer = PGTrainer(args, {"maddpg": MADDPG, "safemaddpg": SAFEMADDPG, "matd3": MATD3, "iddpg": IDDPG}[a.alg], env, None,
                        batch_scale=a.batch_scale, replay_capacity=a.envs * 96 * 2)
    stat = {}
    trainer.behaviour_net.train_process(stat, trainer)          # warm-up episode (allocations, rocBLAS plans)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for ep in range(a.episodes):
        trainer.behaviour_net.train_process(stat, trainer)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    steps = a.episodes * 95
    if rank == 0:
        out = {"metric": "training env-steps/s (rollout + replay + MADDPG updates)", "alg": a.alg,
               "value": a.envs * world * steps / dt, "unit": "env-steps/s", "n_gpus": world, "envs_per_gpu": a.envs,
               "n_agents": env.n_agents, "vector_steps": steps, "ms_per_vector_step": dt / steps * 1e3,
               "batch": trainer.effective_batch_size(),
               "grad_steps": int(steps // args.behaviour_update_freq) * 11,
               "stat": {k: (float(v) if not isinstance(v, float) else v) for k, v in stat.items()}}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
