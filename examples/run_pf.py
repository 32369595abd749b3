#!/usr/bin/env python3
"""BASELINE.json config 1 — the deterministic case of run_pf.py:37-54 (every PQ bus P = 0.1 pu, Q = 0.005 pu; buildings
reduce by 50 %, PV at 0.5*pv_cap, ESS charging at p_ch_max from e_max/2) through the HIP power flow, printed in the
reference's return-dict shape (pf.py:108-113).  `--check` compares with the CPU oracle."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    a = ap.parse_args()
    import numpy as np
    import torch
    import safe_marl_amd  # noqa: F401
    from safe_marl_amd.flex_env import DEFAULT_ENV_ARGS, pf_solve_batch
    from safe_marl_amd.network import build_tables, create_network

    cfg = DEFAULT_ENV_ARGS
    net = create_network(cfg)
    buses = net["bus_numbers"]
    pd = {b: 0 if net["bus_types"][b] == 1 else 0.1 for b in buses}                 # run_pf.py:37-38
    qd = {b: 0 if net["bus_types"][b] == 1 else 0.005 for b in buses}
    blds = cfg["buildings"]
    pred = {b: pd[b] * cfg["max_power_reduction"] for b in blds}                      # run_pf.py:41
    ppv = {b: 0.5 * cfg["pv_cap"] for b in cfg["pv_nodes"]}                           # run_pf.py:44
    qpv = {b: 0 for b in cfg["pv_nodes"]}
    ch = {b: cfg["p_ch_max"] for b in cfg["ess_nodes"]}                               # run_pf.py:50-51
    dis = {b: 0 for b in cfg["ess_nodes"]}
    e0 = {b: cfg["e_max"] / 2 for b in cfg["ess_nodes"]}
    pnet = np.array([pd[b] - pred.get(b, 0) - ppv.get(b, 0) + ch.get(b, 0) - dis.get(b, 0) for b in buses])   # pf.py:69-73
    qnet = np.array([qd[b] - qpv.get(b, 0) for b in buses])                                                   # pf.py:81-82
    out = pf_solve_batch(net, torch.from_numpy(pnet[None]).cuda(), torch.from_numpy(qnet[None]).cuda(), want_branch=True)
    t = build_tables(net)
    v = out["v"][0].cpu().numpy()
    res = {
        "Voltages": {b: float(v[i]) for i, b in enumerate(buses)},
        "Currents": {t.line_of_bus[i]: float(np.sqrt(out["isqr"][0, i].item())) for i in range(t.n_bus) if t.line_of_bus[i]},
        "Power Flows": {t.line_of_bus[i]: (float(out["pl"][0, i]), float(out["ql"][0, i])) for i in range(t.n_bus) if t.line_of_bus[i]},
        "Next ESS Energy": {b: e0[b] + (24 / cfg["episode_limit"]) * (cfg["eta_ch"] * ch[b] - dis[b] / cfg["eta_dis"]) for b in blds},
    }
    print("Power flow results (stand-in IEEE-33 Baran-Wu network, not the reference's xlsx):")
    print("  min |V| = %.6f pu at bus %d" % (v.min(), buses[int(v.argmin())]))
    print("  Voltages:", {k: round(x, 6) for k, x in res["Voltages"].items()})
    print("  Next ESS Energy:", res["Next ESS Energy"])
    if a.check:
        from oracle import pf_oracle
        ref = pf_oracle.power_flow_solver(net, pd, qd, pred, ppv, qpv, ch, dis, e0)
        dv = max(abs(ref["Voltages"][b] - res["Voltages"][b]) for b in buses)
        di = max(abs(ref["Currents"][k] - res["Currents"][k]) for k in ref["Currents"])
        print(f"  vs CPU oracle: |dV| <= {dv:.2e}, |dI| <= {di:.2e}")
        assert dv < 1e-10 and di < 1e-9


if __name__ == "__main__":
    main()
