#!/usr/bin/env python3
"""The flow of the reference's run_opf.py (:34-75) on the device: take one day's record — the test record a
``PGTester`` run pickled (keys pv_active, bus_active, bus_reactive, ess_energy, price: run_opf.py:38-46), or a day
of the synthetic series — build the reference's input dicts (run_opf.py:51-69) and call ``opf_model``.
`--days N` instead solves N days at once through the tensor interface and prints the spread of the optimum."""
import argparse
import os
import pickle
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--record", help="pickle written by PGTester (tester.py:23-33 keys)")
    ap.add_argument("--day", type=int, default=5)
    ap.add_argument("--days", type=int, default=0, help="solve this many consecutive days in one batch")
    a = ap.parse_args()
    import numpy as np
    import safe_marl_amd  # noqa: F401
    from safe_marl_amd.flex_env import DEFAULT_ENV_ARGS
    from safe_marl_amd.network import create_network
    from safe_marl_amd.opf import BatchedOPF, opf_model
    from safe_marl_amd.series import make_synthetic_series

    cfg = DEFAULT_ENV_ARGS
    net = create_network(cfg)
    T = cfg["episode_limit"]
    buses = net["bus_numbers"]
    if a.days:
        tab = np.asarray(make_synthetic_series(net, n_days=a.day + a.days + 1).table)
        rows = np.stack([tab[96 * (a.day + b):96 * (a.day + b) + T] for b in range(a.days)])
        r = BatchedOPF(net, cfg).solve(rows[:, :, 71], rows[:, :, :33], rows[:, :, 33:66], rows[:, :, 66:71],
                                       np.full((a.days, len(net["buildings"])), cfg["e_max"] / 2))
        obj = r["objective"].cpu().numpy()
        print(f"{a.days} days: objective mean {obj.mean():+.6f}, min {obj.min():+.6f}, max {obj.max():+.6f}; "
              f"min |V| {float(r['Vsqr'].min().sqrt()):.4f} pu; {r['outer_iters']} outer iterations")
        return
    if a.record:
        with open(a.record, "rb") as f:
            rec = pickle.load(f)
        pv_active, bus_active, bus_reactive = (np.asarray(rec[k]) for k in ("pv_active", "bus_active", "bus_reactive"))
        ess_energy, price = np.asarray(rec["ess_energy"]), np.asarray(rec["price"]).reshape(len(rec["price"]), -1)[:, 0]
    else:
        tab = np.asarray(make_synthetic_series(net, n_days=a.day + 2).table)[96 * a.day:96 * a.day + T]
        bus_active, bus_reactive, pv_active, price = tab[:, :33], tab[:, 33:66], tab[:, 66:71], tab[:, 71]
        ess_energy = np.full((T, len(net["ESSs_at_buildings"])), cfg["e_max"] / 2)
    flex_price = {t + 1: float(price[t]) for t in range(T)}                                         # run_opf.py:51
    active = {n: [float(bus_active[t][i]) for t in range(T)] for i, n in enumerate(buses)}           # run_opf.py:54-57
    reactive = {n: [float(bus_reactive[t][i]) for t in range(T)] for i, n in enumerate(buses)}
    pv = {g: [float(pv_active[t][i]) for t in range(T)] for i, g in enumerate(net["PVs_at_buildings"])}
    e_init = {k: float(ess_energy[0][i]) for i, k in enumerate(net["ESSs_at_buildings"])}          # run_opf.py:69
    sol = opf_model(net, flex_price, active, reactive, pv, e_init, env_args=cfg)
    dt = 24.0 / T
    value = sum(dt * (sum(flex_price[t] * p - cfg["discomfort_coeff"] * p * p for p in sol["Power Reduction"][t].values())
                      - cfg["pv_cost"] * sum(sol["PV Reactive Power"][t].values())
                      - cfg["ess_cost"] * (sum(sol["ESS Charging"][t].values()) + sum(sol["ESS Discharging"][t].values()))
                      - sum(net["line_resistances"][k] * i2 for k, i2 in sol["Current Squared"][t].items()))
                for t in range(1, T + 1))
    vmin = min(min(v.values()) for v in sol["Voltage Squared"].values()) ** 0.5
    print("Optimization results (stand-in IEEE-33 Baran-Wu network, not the reference's xlsx):")
    print(f"  objective (opf.py:80-93) = {value:+.6f}; min |V| = {vmin:.4f} pu")
    for t in (1, T // 2, T):
        print(f"  t={t:2d}: Pred {[round(v, 4) for v in sol['Power Reduction'][t].values()]}  "
              f"Qpv {[round(v, 4) for v in sol['PV Reactive Power'][t].values()]}  "
              f"E {[round(v, 5) for v in sol['ESS Energy'][t].values()]}")


if __name__ == "__main__":
    main()
