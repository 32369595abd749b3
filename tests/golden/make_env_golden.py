#!/usr/bin/env python3
"""Generates tests/golden/env_golden.npz by IMPORTING AND RUNNING the reference's own environment code —
madrl/environments/flex_provision/flexibility_provision_env.py (reset, manual_reset, step, get_obs, get_state,
calculate_reward, action scaling / clipping, ESS clipping, data-row handling, CSV resampling) and utils/create_net.py
(per-unit scaling) — on seeded synthetic data, and recording inputs and outputs.  Run in the build container only:

    cd /root/reference && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/make_env_golden.py

THREE SUBSTITUTIONS, each forced by what this container lacks (ordinary missing dependencies, nothing was
refused), each confined to this script, none touching the reference's files:

 (1) THE POWER-FLOW SOLVE.  utils/pf.py imports pyomo (absent) and hands its NLP to an IPOPT binary (absent).  Empty
     placeholder modules named pyomo / pyomo.environ / pyomo.opt let `import utils.pf` execute, and the NAME
     `power_flow_solver` inside the environment module is rebound to oracle/pf_oracle.power_flow_solver (same signature,
     same result dict; polar Newton-Raphson on the Ybus — the solver tests/ already check against the reference's NLP
     statement, oracle/pf_nlp_oracle.py).  => THE VOLTAGES IN THESE FIXTURES ARE THE ORACLE'S, NOT IPOPT'S.  Everything
     the environment does around the solve — which inputs it hands over, what it does with voltages and ESS energies,
     rewards, observations, the data-row lag, roll-back on failure — is the reference's code, executed here.
 (2) THE WORKBOOKS.  pandas.read_excel has no engine here (openpyxl absent) and Nodes_33.xlsx / Lines_33.xlsx are Git-LFS
     pointers: pandas.read_excel is rebound to the package's own xlsx reader (safe_marl_amd.network.read_xlsx_table) over
     workbooks written from the stand-in IEEE-33 tables.  create_network() itself is the reference's.
 (3) THE TIME SERIES.  The four CSVs are Git-LFS pointers: synthetic ones are written in the reference's layout (a `time`
     column + 32 / 32 / 5 / 1 value columns, 3-minute rows) into a temporary directory handed over as `data_path`.

What the fixtures therefore pin: SURVEY §8 rows a3-a12 (everything but the numerical solve a1/a2) against the reference's
executed code, for the scaled-action branch and the `safemaddpg` raw branch, over whole episodes, a second reset, a
manual_reset and an injected solver failure.  No reference source text travels: the fixture is arrays.
"""
import contextlib
import io
import os
import sys
import tempfile
import types

import numpy as np
import pandas as pd

REF = "/root/reference"
OUT = os.environ.get("GOLDEN_OUT", os.path.dirname(os.path.abspath(__file__)))
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REF)
sys.path.insert(0, REPO)
os.chdir(REF)

# ---- substitution (1a): placeholders so that `import pyomo.environ as pyo` / `from pyomo.opt import SolverFactory` execute
for name in ("pyomo", "pyomo.environ", "pyomo.opt"):
    sys.modules.setdefault(name, types.ModuleType(name))
sys.modules["pyomo.opt"].SolverFactory = None

import safe_marl_amd  # noqa: E402,F401
from safe_marl_amd.network import ieee33_tables, read_xlsx_table  # noqa: E402
from oracle import pf_oracle  # noqa: E402

TMP = tempfile.mkdtemp(prefix="env_golden_")


def write_inputs():
    """Workbooks (stand-in IEEE-33) and CSVs (synthetic, 3-minute rows over 14 days) in the reference's on-disk formats."""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from conftest import _write_xlsx
    nodes, lines = ieee33_tables()
    _write_xlsx(os.path.join(TMP, "Nodes_33.xlsx"), ["NODES", "Tb", "PDn", "QDn"],
                [(float(n[0]), float(n[1]), float(n[2]), float(n[3])) for n in nodes])
    _write_xlsx(os.path.join(TMP, "Lines_33.xlsx"), ["FROM", "TO", "R", "X", "Imax"],
                [(float(l[0]), float(l[1]), float(l[2]), float(l[3]), float(l[4])) for l in lines])
    rng = np.random.default_rng(20250114)
    idx = pd.date_range("2021-03-01", periods=14 * 24 * 20, freq="3min")
    hours = (idx.hour + idx.minute / 60.0).to_numpy()
    pbase = np.array([n[2] for n in nodes[1:]]) / 1000.0          # pu on s_nom = 1000 kVA
    qbase = np.array([n[3] for n in nodes[1:]]) / 1000.0
    shape = (0.6 + 0.4 * np.sin(np.pi * hours / 24.0) ** 2)[:, None]
    active = pbase[None] * shape * rng.uniform(0.8, 1.2, (len(idx), 32))
    reactive = qbase[None] * shape * rng.uniform(0.8, 1.2, (len(idx), 32))
    pv = np.maximum(0.0, np.sin(np.pi * (hours - 6.0) / 12.0))[:, None] * rng.uniform(0.7, 1.0, (len(idx), 5))   # x pv_scale in the env
    price = rng.uniform(0.05, 0.30, (len(idx), 1))
    for name, arr in (("load_active.csv", active), ("load_reactive.csv", reactive), ("pv_active.csv", pv), ("prices.csv", price)):
        df = pd.DataFrame(arr, columns=[f"c{i}" for i in range(arr.shape[1])])
        df.insert(0, "time", idx)
        df.to_csv(os.path.join(TMP, name), index=False)


def fake_read_excel(path, *a, **k):
    """substitution (2): the package's xlsx reader on the workbook of the same name in TMP"""
    header, rows = read_xlsx_table(os.path.join(TMP, os.path.basename(str(path))))
    df = pd.DataFrame(rows, columns=header)
    for c in ("NODES", "Tb", "FROM", "TO"):
        if c in df:
            df[c] = df[c].astype(int)
    return df


def main():
    write_inputs()
    pd.read_excel = fake_read_excel
    import yaml
    with open("madrl/args/env_args/flex_provision.yaml") as f:
        env_args = yaml.safe_load(f)["env_args"]
    from madrl.environments.flex_provision import flexibility_provision_env as E
    fail_at = {"calls": None}
    calls = {"n": 0}

    def oracle_pf(*args):                                       # substitution (1b)
        calls["n"] += 1
        if fail_at["calls"] is not None and calls["n"] == fail_at["calls"]:
            raise Exception("Solver failed to find a solution")    # pf.py:104-105's exception, injected (scenario D)
        return pf_oracle.power_flow_solver(*args, env_config=env_args)

    E.power_flow_solver = oracle_pf
    # record the initial action draw of reset / manual_reset (a local there) without changing it
    orig_get_action = E.FlexibilityProvisionEnv.get_action

    def get_action(self):
        a = orig_get_action(self)
        self._golden_a0 = np.array(a, dtype=np.float64).copy()
        return a

    E.FlexibilityProvisionEnv.get_action = get_action

    g = {"note": np.array("reference env code executed with (1) power_flow_solver := oracle/pf_oracle NR, (2) read_excel := package "
                          "xlsx reader on stand-in IEEE-33 workbooks, (3) synthetic CSVs; see make_env_golden.py")}
    buses = None

    def snapshot(env):
        nonlocal buses
        buses = list(env.base_powergrid["bus_numbers"])
        ess = env.base_powergrid["ESSs_at_buildings"]
        return dict(V=np.array([env.current_voltage[b] for b in buses]), E=np.array([env.current_ess_energy[k] for k in ess]),
                    Einit=np.array([env.initial_ess_energy[k] for k in ess]), steps=env.steps, cum=env.cumulative_reward)

    def run(tag, alg, seed, n_steps, lo, hi, second_reset=False, manual=None, fail_step=None):
        kw = dict(env_args)
        kw.update(data_path=TMP, seed=seed)
        if alg:
            kw["alg"] = alg
        fail_at["calls"] = None
        calls["n"] = 0
        with contextlib.redirect_stdout(io.StringIO()):
            env = E.FlexibilityProvisionEnv(kw)                 # np.random.seed(seed); reset()
        episodes = []

        def episode(first_obs_state):
            obs0, state0 = first_obs_state
            rec = dict(day=env.start_day, hour=env.start_hour, interval=env.start_interval,
                       e0=np.array([env.initial_ess_energy[k] for k in env.base_powergrid["ESSs_at_buildings"]]),
                       a0=env._golden_a0, obs=[np.stack(obs0)], state=[state0], snaps=[snapshot(env)], actions=[], reward=[],
                       done=[], info=[], failed=[])
            arng = np.random.default_rng(1000 + seed)
            for t in range(n_steps):
                act = arng.uniform(lo, hi, 20).astype(np.float32).astype(np.float64)      # float32 values (util.py:184)
                if fail_step is not None and t == fail_step:
                    fail_at["calls"] = calls["n"] + 1
                with contextlib.redirect_stdout(io.StringIO()):
                    r, d, info = env.step(act.copy())
                    ob = env.get_obs()                          # model.py:223: get_obs() right after step()
                rec["actions"].append(act); rec["reward"].append(r); rec["done"].append(d)
                rec["failed"].append(bool(info.get("solver_failed", False)))
                rec["info"].append([info[k] for k in ("reward", "revenue", "der_cost", "ess_cost", "discomfort_penalty",
                                                      "voltage_penalty", "cumulative_reward")])
                rec["obs"].append(np.stack(ob)); rec["state"].append(env.get_state()); rec["snaps"].append(snapshot(env))
                if d:
                    break
            return rec

        # the constructor's reset() already happened: its obs/state come from a get_obs() inside reset — re-read them
        first = None
        # (reset() returned them to the constructor only; reproduce by the documented sequence: a fresh reset)
        np.random.seed(seed)
        with contextlib.redirect_stdout(io.StringIO()):
            first = env.reset()
        episodes.append(episode(first))
        if second_reset:
            with contextlib.redirect_stdout(io.StringIO()):
                first = env.reset()                              # the global NumPy stream simply continues
            episodes.append(episode(first))
        if manual is not None:
            with contextlib.redirect_stdout(io.StringIO()):
                first = env.manual_reset(*manual)
            episodes.append(episode(first))
        for j, rec in enumerate(episodes):
            p = f"{tag}.ep{j}."
            g[p + "start"] = np.array([rec["day"], rec["hour"], rec["interval"]], np.int64)
            g[p + "e0"], g[p + "a0"] = rec["e0"], rec["a0"]
            g[p + "actions"] = np.array(rec["actions"]); g[p + "reward"] = np.array(rec["reward"])
            g[p + "done"] = np.array(rec["done"]); g[p + "failed"] = np.array(rec["failed"])
            g[p + "info"] = np.array(rec["info"], np.float64)
            g[p + "obs"] = np.array(rec["obs"]); g[p + "state"] = np.array(rec["state"])
            g[p + "V"] = np.array([s["V"] for s in rec["snaps"]]); g[p + "E"] = np.array([s["E"] for s in rec["snaps"]])
            g[p + "Einit"] = np.array([s["Einit"] for s in rec["snaps"]])
            g[p + "steps"] = np.array([s["steps"] for s in rec["snaps"]]); g[p + "cum"] = np.array([s["cum"] for s in rec["snaps"]])
        g[tag + ".seed"], g[tag + ".alg"], g[tag + ".episodes"] = np.array(seed), np.array(alg or ""), np.array(len(episodes))
        return env

    env = run("A", None, 0, 95, 0.0, 1.0, second_reset=True)                       # full episode + a second reset, ESS clipping branches
    run("B", None, 3, 95, 0.5, 1.0, manual=(4, 13, 2))                             # the range the policy path delivers (A1) + manual_reset
    run("C", "safemaddpg", 5, 40, 0.0, 1.0)                                        # raw-action branch (env:268-274)
    run("D", None, 7, 20, 0.5, 1.0, fail_step=6)                                   # injected solver failure at step 7 (A8)

    # the resampled series the env built from the CSVs (env:431-471) and the reference's create_network() dict
    g["series.active"] = env.active_demand_data.to_numpy(); g["series.reactive"] = env.reactive_demand_data.to_numpy()
    g["series.pv"] = env.pv_data.to_numpy(); g["series.price"] = env.price_data.to_numpy()
    g["series.time_delta"] = np.array(env.time_delta)
    net = env.base_powergrid
    g["net.bus_numbers"] = np.array(net["bus_numbers"], np.int64)
    lines = sorted(net["line_connections"])
    g["net.lines"] = np.array(lines, np.int64)
    g["net.r"] = np.array([net["line_resistances"][l] for l in lines]); g["net.x"] = np.array([net["line_reactances"][l] for l in lines])
    g["net.imax"] = np.array([net["max_line_currents"][l] for l in lines])
    g["net.types"] = np.array([net["bus_types"][b] for b in net["bus_numbers"]], np.int64)
    g["net.pd"] = np.array([net["active_power_demand"][b] for b in net["bus_numbers"]])
    g["net.qd"] = np.array([net["reactive_power_demand"][b] for b in net["bus_numbers"]])
    g["net.buildings"] = np.array(net["buildings"], np.int64)
    import json
    g["env_args_json"] = np.array(json.dumps({k: v for k, v in env_args.items()}, sort_keys=True))
    np.savez_compressed(os.path.join(OUT, "env_golden.npz"), **g)
    print("wrote env_golden.npz:", len(g), "arrays;", {k: g[k].shape for k in ("A.ep0.obs", "A.ep1.reward", "B.ep1.obs", "C.ep0.reward", "D.ep0.failed")})
    print("D failed flags:", g["D.ep0.failed"].astype(int).tolist(), "rewards", np.round(g["D.ep0.reward"], 3).tolist())


if __name__ == "__main__":
    main()
