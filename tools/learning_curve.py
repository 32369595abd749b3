#!/usr/bin/env python3
"""Does the vectorised trainer LEARN?  

The reference's product is a trained policy: 400 training episodes (train_agent.py:125-128) with a test-mode evaluation
every 20 (utils/trainer.py:120-124, madrl/models/model.py:269-306).  This tool runs that schedule on the HIP path and
records the evaluation curve of every run on ONE fixed set of evaluation episodes (injected reset draws: day, hour,
interval, E0, initial action — every policy sees the same episodes), next to

  (i)   the untrained policy (evaluation 0 of every run),
  (ii)  the N = 1 reference-cadence path (FlexibilityProvisionEnv, batch 32, update per 60 env steps) for the same
        number of gradient steps,
  (iii) the f4 OPF comparator (utils/opf.py via safe_marl_amd.opf.BatchedOPF) on a subset of the same episodes,
        evaluated in the environment's own reward terms,

at ``batch_scale`` = n_envs / 4 (1.47 samples consumed per transition collected) and n_envs (5.87, the reference's
ratio: 11 gradient steps x 32 samples per 60 transitions, model.py:43-50 x replay_buffer.py:17-21).

    python tools/learning_curve.py --out profiles/r03_learning_curve.json            # everything (~2-4 min of GPU)
    python tools/learning_curve.py --episodes 40 --runs maddpg:4096:4 --no-ref --no-opf   # a quick look

Every number is on the stand-in IEEE-33 feeder and the synthetic series, not the reference's data.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "examples"))

TERMS = ("reward", "revenue", "der_cost", "ess_cost", "discomfort_penalty", "voltage_penalty", "violation_rate",
         "solver_failed")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--episodes", type=int, default=400, help="training episodes per run (train_agent.py:125: 400)")
    ap.add_argument("--eval-freq", type=int, default=20, help="evaluate every this many episodes (default.yaml: eval_freq)")
    ap.add_argument("--eval-envs", type=int, default=1024, help="fixed evaluation episodes")
    ap.add_argument("--runs", default="maddpg:4096:4,maddpg:4096:1,safemaddpg:8192:4,safemaddpg:8192:1",
                    help="alg:envs:d[:intended] per run; batch_scale = envs / d; 'intended' = SAFEMADDPG.intended_actions (the "
                         "safety layer's output reaches the env as physical values, NOT the reference's routing, SURVEY A13)")
    ap.add_argument("--load-scale", type=float, default=1.0,
                    help="multiply every bus's active and reactive demand by this (a stressed feeder: under-voltages appear "
                         "from about 1.2; the generated series never leaves [0.9, 1.1] pu at 1.0)")
    ap.add_argument("--no-ref", action="store_true", help="skip the N = 1 reference-cadence run")
    ap.add_argument("--ref-episodes", type=int, default=None, help="episodes of the N = 1 run (default: --episodes)")
    ap.add_argument("--no-opf", action="store_true")
    ap.add_argument("--opf-days", type=int, default=32, help="evaluation episodes handed to the OPF comparator")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--lr", type=float, default=None, help="override policy_lrate / value_lrate (default.yaml: 1e-4)")
    ap.add_argument("--out", default=None)
    return ap.parse_args()


def eval_spec(series, n, n_agents, seed=2025):
    """The fixed evaluation episodes: the draws of env:85-87,100,103, from a generator of their own."""
    import numpy as np
    rng = np.random.default_rng(seed)
    return dict(day=rng.integers(0, series.n_start_days(96), n).astype(np.int32),
                hour=rng.integers(0, 24, n).astype(np.int32), interval=rng.integers(0, 4, n).astype(np.int32),
                e0=rng.uniform(0.9 * 0.0125, 1.1 * 0.0125, (n, n_agents)), a0=rng.uniform(0.0, 1.0, (n, 4 * n_agents)))


def strip(stat):
    return {k: float(stat["mean_test_" + k]) for k in TERMS}


def main():
    a = parse()
    import numpy as np
    import torch
    import safe_marl_amd  # noqa: F401
    from train_maddpg import DEFAULT_ALG_ARGS
    from safe_marl_amd import learner
    from safe_marl_amd.flex_env import FlexibilityProvisionEnv, VecFlexProvisionEnv
    from safe_marl_amd.network import create_network
    from safe_marl_amd.series import make_synthetic_series
    from safe_marl_amd.trainer import PGTrainer
    from safe_marl_amd.util import convert

    classes = {"maddpg": learner.MADDPG, "safemaddpg": learner.SAFEMADDPG, "matd3": learner.MATD3, "iddpg": learner.IDDPG}
    net = create_network({})
    series = make_synthetic_series(net, n_days=365)
    if a.load_scale != 1.0:
        series.table[:, :2 * len(net["bus_numbers"])] *= a.load_scale
    n_agents = len(net["buildings"])
    spec = eval_spec(series, a.eval_envs, n_agents)
    # one evaluation env per action convention (env:268: safemaddpg hands its actions over raw), same episodes
    eval_envs = {}

    def eval_env(alg):
        key = "safemaddpg" if alg == "safemaddpg" else "plain"
        if key not in eval_envs:
            env_args = {"alg": "safemaddpg"} if key == "safemaddpg" else {}
            eval_envs[key] = VecFlexProvisionEnv(env_args, a.eval_envs, net=net, series=series, seed=77, warm_start=True)
        return eval_envs[key]

    def alg_args(alg, env):
        d = dict(DEFAULT_ALG_ARGS)
        d.update(alg=alg, agent_num=env.n_agents, obs_size=env.obs_size, state_size=env.state_size, action_dim=4,
                 v_min=0.9, v_max=1.1)
        if a.lr is not None:
            d.update(policy_lrate=a.lr, value_lrate=a.lr)
        return convert(d)

    out = {"what": "evaluation curves on fixed episodes: mean per env-step over %d episodes x 95 steps, test mode "
                   "(tanh(mean), util.py:79-82)" % a.eval_envs,
           "data": "synthetic (stand-in IEEE-33 Baran-Wu network, generated series; SURVEY.md App. C, §8d)",
           "load_scale": a.load_scale, "episodes": a.episodes, "eval_freq": a.eval_freq, "eval_envs": a.eval_envs, "seed": a.seed,
           "lr": a.lr if a.lr is not None else DEFAULT_ALG_ARGS["policy_lrate"], "runs": []}

    def gradient_steps(args, steps):
        events = sum(1 for st in range(steps) if st > 0 and st % args.behaviour_update_freq == 0)
        return events * (args.value_update_epochs + args.policy_update_epochs)

    for item in [r for r in a.runs.split(",") if r]:
        alg, envs, div, *flags = item.split(":")
        envs, div = int(envs), int(div)
        intended = "intended" in flags
        env_args = {"alg": "safemaddpg"} if alg == "safemaddpg" else {}
        env = VecFlexProvisionEnv(env_args, envs, net=net, series=series, seed=1234, warm_start=True)
        args = alg_args(alg, env)
        torch.manual_seed(a.seed)
        np.random.seed(a.seed)
        tr = PGTrainer(args, classes[alg], env, None, batch_scale=max(1, envs // div), replay_capacity=envs * 96 * 2)
        if intended:
            tr.behaviour_net.intended_actions = True
            alg_label = alg + " (intended action routing, not the reference's)"
        else:
            alg_label = alg
        ev = eval_env(alg)
        bs = tr.effective_batch_size()
        per_event = args.value_update_epochs + args.policy_update_epochs
        run = {"alg": alg_label, "mode": "vectorised", "envs": envs, "batch_scale": tr.batch_scale, "batch": bs,
               "samples_per_transition": per_event * bs / float(args.behaviour_update_freq * envs), "curve": []}
        t0 = time.perf_counter()
        train_s = 0.0
        for ep in range(a.episodes + 1):
            if ep % a.eval_freq == 0 or ep == a.episodes:
                e = strip(tr.behaviour_net.evaluate_on(ev, spec))
                e.update(episode=ep, vector_steps=tr.steps, gradient_steps=gradient_steps(args, tr.steps))
                run["curve"].append(e)
                print(f"[{alg_label} x{envs} bs={bs}] ep {ep:4d} test reward {e['reward']:+.5f} rev {e['revenue']:.5f} der {e['der_cost']:+.5f} "
                      f"ess {e['ess_cost']:.5f} disc {e['discomfort_penalty']:.5f} vpen {e['voltage_penalty']:.5f} "
                      f"viol {e['violation_rate']:.4f}", flush=True)
            if ep == a.episodes:
                break
            stat = {}
            t1 = time.perf_counter()
            tr.behaviour_net.train_process(stat, tr)
            torch.cuda.synchronize()
            train_s += time.perf_counter() - t1
            for k, v in list(stat.items()):
                if isinstance(v, torch.Tensor):
                    stat[k] = float(v.item())
            bad = [k for k, v in stat.items() if not np.isfinite(v)]
            if bad:
                raise SystemExit(f"non-finite statistic {bad} at episode {ep}")
            if ep % a.eval_freq == 0:
                run.setdefault("train", []).append({"episode": ep, **{k[len("mean_train_"):]: float(v) for k, v in stat.items()}})
        run["train_seconds"] = train_s
        run["train_env_steps_per_s"] = envs * tr.steps / train_s
        run["wall_seconds"] = time.perf_counter() - t0
        rg = getattr(tr.behaviour_net, "_rollout_graph", None)
        run["rollout_graph"] = bool(rg is not None and rg.graph is not None)
        run["graphed_updates"] = sorted(tr._update_graphs)
        out["runs"].append(run)
        del tr, env
        torch.cuda.empty_cache()

    if not a.no_ref:
        # (ii) the reference's own cadence: one env, batch 32, 10 value + 1 policy sub-update per 60 env steps
        episodes = a.ref_episodes or a.episodes
        env1 = FlexibilityProvisionEnv({}, net=net, series=series, warm_start=True)
        args = alg_args("maddpg", env1.vec)
        torch.manual_seed(a.seed)
        np.random.seed(a.seed)
        tr = PGTrainer(args, learner.MADDPG, env1, None)
        ev = eval_env("maddpg")
        run = {"alg": "maddpg", "mode": "N=1 reference cadence (FlexibilityProvisionEnv, model.py:198-267)", "envs": 1,
               "batch_scale": 1, "batch": tr.effective_batch_size(),
               "samples_per_transition": 11 * 32 / 60.0, "curve": []}
        t0 = time.perf_counter()
        import contextlib
        import io
        for ep in range(episodes + 1):
            if ep % a.eval_freq == 0 or ep == episodes:
                e = strip(tr.behaviour_net.evaluate_on(ev, spec))
                e.update(episode=ep, vector_steps=tr.steps, gradient_steps=gradient_steps(args, tr.steps))
                run["curve"].append(e)
                print(f"[maddpg N=1] ep {ep:4d} test reward {e['reward']:+.5f} rev {e['revenue']:.5f} der {e['der_cost']:+.5f} "
                      f"ess {e['ess_cost']:.5f} disc {e['discomfort_penalty']:.5f} vpen {e['voltage_penalty']:.5f} "
                      f"viol {e['violation_rate']:.4f}  ({time.perf_counter() - t0:.0f} s)", flush=True)
            if ep == episodes:
                break
            stat = {}
            with contextlib.redirect_stdout(io.StringIO()):       # env:351 prints a line per episode
                tr.behaviour_net.train_process(stat, tr)
        run["wall_seconds"] = time.perf_counter() - t0
        out["runs"].append(run)
        del tr

    if not a.no_opf:
        # (iii) opf.py on the first --opf-days evaluation episodes: the rows the episode steps through (row 1 .. 95 of the
        # slice), E0 as drawn; valued in the env's reward terms per step (env:679-706), voltage limits hard
        from safe_marl_amd.opf import BatchedOPF
        nd = min(a.opf_days, a.eval_envs)
        tab = np.asarray(series.table)
        start = spec["interval"][:nd] + spec["hour"][:nd] * 4 + spec["day"][:nd] * 96
        T = 95
        rows = np.stack([tab[s + 1:s + 1 + T] for s in start])
        nb = len(net["bus_numbers"])
        cfg = dict(eval_env("maddpg").args_dict)
        t0 = time.perf_counter()
        opf = BatchedOPF(net, cfg)
        ok, terms = [], []
        for lo in range(0, nd, 16):
            hi = min(nd, lo + 16)
            sl = slice(lo, hi)
            try:
                r = opf.solve(rows[sl, :, -1], rows[sl, :, :nb], rows[sl, :, nb:2 * nb], rows[sl, :, 2 * nb:2 * nb + n_agents],
                              spec["e0"][sl])
            except Exception as exc:                      # an infeasible day in the block: one day at a time
                print(f"[opf] block {lo}:{hi}: {exc}; solving its days one by one", flush=True)
                for d in range(lo, hi):
                    try:
                        r1 = opf.solve(rows[d:d + 1, :, -1], rows[d:d + 1, :, :nb], rows[d:d + 1, :, nb:2 * nb],
                                       rows[d:d + 1, :, 2 * nb:2 * nb + n_agents], spec["e0"][d:d + 1])
                        ok.append(d)
                        terms.append({k: r1[k].cpu().numpy() for k in ("Pred", "Qpv", "Pesc", "Pesd", "Vsqr")})
                    except Exception:
                        pass
                continue
            ok.extend(range(lo, hi))
            terms.append({k: r[k].cpu().numpy() for k in ("Pred", "Qpv", "Pesc", "Pesd", "Vsqr")})
        if ok:
            cat = {k: np.concatenate([t[k] for t in terms]) for k in terms[0]}
            price = rows[ok, :, -1]
            rev = (price[:, :, None] * cat["Pred"]).sum(-1)
            der = cfg["pv_cost"] * cat["Qpv"].sum(-1)
            ess = cfg["ess_cost"] * (cat["Pesc"] + cat["Pesd"]).sum(-1)
            disc = cfg["discomfort_coeff"] * (cat["Pred"] ** 2).sum(-1)
            v = np.sqrt(cat["Vsqr"])
            vpen = cfg["voltage_coeff"] * np.maximum(0, np.maximum(v - cfg["v_max"], cfg["v_min"] - v)).sum(-1)
            out["opf"] = {"episodes": len(ok), "of": nd, "seconds": time.perf_counter() - t0,
                          "what": "utils/opf.py:13-192 (BatchedOPF) on the evaluation episodes' rows, valued per step in the "
                                  "env's reward terms (env:679-706); its controls range over [0, max] where translate_action "
                                  "(util.py:125-128) confines a policy to the upper half (SURVEY A1): an upper bound, not a target",
                          "reward": float((rev - der - ess - disc - vpen).mean()), "revenue": float(rev.mean()),
                          "der_cost": float(der.mean()), "ess_cost": float(ess.mean()), "discomfort_penalty": float(disc.mean()),
                          "voltage_penalty": float(vpen.mean()), "violation_rate": float((vpen > 1e-9).mean())}
            # the same episodes under every trained policy, for a like-for-like gap
            print(f"[opf] {len(ok)}/{nd} episodes: reward {out['opf']['reward']:+.5f} rev {out['opf']['revenue']:.5f} "
                  f"der {out['opf']['der_cost']:+.5f} ess {out['opf']['ess_cost']:.5f} disc {out['opf']['discomfort_penalty']:.5f}",
                  flush=True)

    # summary table
    lines = ["| run | batch | samples/transition | test reward: untrained -> trained (best) | violation rate: untrained -> trained | train env-steps/s |",
             "|---|---|---|---|---|---|"]
    for r in out["runs"]:
        c = r["curve"]
        best = max(c, key=lambda e: e["reward"])
        lines.append(f"| {r['alg']} {r['mode'] if r['envs'] == 1 else str(r['envs']) + ' envs'} | {r['batch']} | "
                     f"{r['samples_per_transition']:.2f} | {c[0]['reward']:+.5f} -> {c[-1]['reward']:+.5f} ({best['reward']:+.5f} @ {best['episode']}) | "
                     f"{c[0]['violation_rate']:.4f} -> {c[-1]['violation_rate']:.4f} | "
                     f"{r.get('train_env_steps_per_s', float('nan')) / 1e6:.1f} M |")
    if "opf" in out:
        lines.append(f"| OPF comparator ({out['opf']['episodes']} episodes) | - | - | {out['opf']['reward']:+.5f} | "
                     f"{out['opf']['violation_rate']:.4f} | - |")
    out["summary_md"] = lines
    print("\n".join(lines), flush=True)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        with open(a.out, "w") as f:
            json.dump(out, f, indent=1)
        print("wrote", a.out)


if __name__ == "__main__":
    main()
