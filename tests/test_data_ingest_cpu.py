"""CPU: the reference's data formats — CSV ingestion with the 15-min resample of env:467-471,
and create_network's per-unit scaling (create_net.py:17-24)."""
import numpy as np
import pytest

from safe_marl_amd.network import build_tables, create_network, ieee33_tables
from safe_marl_amd.series import load_csv_dir, make_synthetic_series


def test_create_network_per_unit_scaling():
    net = create_network()
    z_base = 12.66 ** 2 * 1000 / 1000
    assert net["line_resistances"][(1, 2)] == pytest.approx(0.0922 / z_base)
    assert net["line_reactances"][(32, 33)] == pytest.approx(0.5302 / z_base)
    assert net["active_power_demand"][24] == pytest.approx(0.42) and net["reactive_power_demand"][30] == pytest.approx(0.6)
    assert net["max_line_currents"][(1, 2)] == pytest.approx(400.0 / (1000 / 12.66))
    assert net["bus_types"][1] == 1 and sum(net["bus_types"].values()) == 1
    assert net["buildings"] == [5, 10, 15, 20, 25]
    t = build_tables(net)
    assert t.n_levels == 18 and t.max_children == 2 and list(t.agent_bus) == [4, 9, 14, 19, 24]
    assert t.parent[18] == 1 and t.parent[22] == 2 and t.parent[25] == 5      # laterals at buses 2, 3, 6


def test_rejects_meshed_or_disconnected_networks():
    nodes, lines = ieee33_tables()
    with pytest.raises(ValueError):
        build_tables(create_network(None, nodes, lines + [(8, 21, 2.0, 2.0, 400.0)]))      # a tie-line: not radial
    bad = [l for l in lines if l[:2] != (6, 26)] + [(27, 26, 0.2030, 0.1034, 400.0)]
    with pytest.raises(ValueError):
        build_tables(create_network(None, nodes, bad))                                    # lateral cut off the slack


def test_csv_ingestion_matches_reference_resampling(tmp_path, net):
    pd = pytest.importorskip("pandas")
    rng = np.random.default_rng(0)
    idx = pd.date_range("2020-01-01", periods=5 * 24 * 20, freq="3min")          # 3-min raw data, like the reference's
    for name, cols in (("load_active.csv", 32), ("load_reactive.csv", 32), ("pv_active.csv", 5), ("prices.csv", 1)):
        df = pd.DataFrame(rng.uniform(0, 1, (len(idx), cols)), columns=[f"c{i}" for i in range(cols)])
        df.insert(0, "time", idx)
        if name == "prices.csv":
            df.iloc[7:9, 1] = np.nan                                              # gaps are interpolated (env:470)
        df.to_csv(tmp_path / name, index=False)
    st = load_csv_dir(net, str(tmp_path), {"pv_scale": 0.15})
    assert st.time_delta == 15 and st.n_bus == 33 and st.n_agents == 5 and st.cols == 72
    assert st.rows == len(idx) // 5
    raw = pd.read_csv(tmp_path / "pv_active.csv")
    assert st.pv[0, 2] == pytest.approx(0.15 * raw.iloc[:5, 3].mean())
    assert (st.active[:, 0] == 0).all() and (st.reactive[:, 0] == 0).all()        # slack column, env:489-490
    assert np.isfinite(st.table).all()


def test_synthetic_series_layout(net):
    st = make_synthetic_series(net, n_days=3)
    assert st.table.shape == (288, 72) and st.pv_days == 2 and st.per_hour == 4
    assert (st.pv[:24] == 0).all() and st.pv[48:72].max() > 0.05                  # no PV before 06:00
    assert st.price.min() >= 0.05 and st.price.max() <= 0.30
