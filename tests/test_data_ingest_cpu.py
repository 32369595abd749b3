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


@pytest.mark.parametrize("shared_strings", [True, False])
def test_network_xlsx_ingestion(tmp_path, shared_strings, write_xlsx):
    """utils/create_net.py:11-24 on the reference's on-disk format: Nodes_33.xlsx (NODES, Tb, PDn, QDn) and
    Lines_33.xlsx (FROM, TO, R, X, Imax) -> the same dict as create_network() on the same tables, and from there the
    same kernel tables.  (The reference's own files are Git-LFS pointers; the workbooks are written here.)"""
    from safe_marl_amd.network import load_network_xlsx, read_xlsx_table
    nodes, lines = ieee33_tables()
    # extra / reordered columns, as real workbooks have them
    write_xlsx(tmp_path / "Nodes_33.xlsx", ["NODES", "Tb", "PDn", "QDn", "comment"],
                [(float(n[0]), float(n[1]), float(n[2]), float(n[3]), 0.0) for n in nodes], shared_strings)
    write_xlsx(tmp_path / "Lines_33.xlsx", ["FROM", "TO", "X", "R", "Imax"],
                [(float(l[0]), float(l[1]), float(l[3]), float(l[2]), float(l[4])) for l in lines], shared_strings)
    header, rows = read_xlsx_table(str(tmp_path / "Lines_33.xlsx"))
    assert header == ["FROM", "TO", "X", "R", "Imax"] and len(rows) == 32 and rows[0][:2] == [1.0, 2.0]
    three = {"buildings": [5, 15, 25], "pv_nodes": [5, 15, 25], "ess_nodes": [5, 15, 25]}
    got = load_network_xlsx(str(tmp_path), three)
    want = create_network(three)
    assert got["bus_numbers"] == want["bus_numbers"] and got["line_connections"] == want["line_connections"]
    for k in ("line_resistances", "line_reactances", "max_line_currents", "bus_types", "active_power_demand",
              "reactive_power_demand"):
        assert got[k] == want[k], k
    assert got["buildings"] == [5, 15, 25]
    a, b = build_tables(got), build_tables(want)
    assert np.array_equal(a.parent, b.parent) and np.array_equal(a.r, b.r) and np.array_equal(a.x, b.x)
    with pytest.raises(KeyError):
        write_xlsx(tmp_path / "Nodes_33.xlsx", ["NODES", "Tb", "PDn"], [(1.0, 1.0, 0.0)], shared_strings)
        load_network_xlsx(str(tmp_path))


def test_xlsx_reader_agrees_with_pandas_writer_when_available(tmp_path):
    """If an xlsx engine happens to be installed, a workbook written by pandas reads back identically."""
    pd = pytest.importorskip("pandas")
    pytest.importorskip("openpyxl")
    from safe_marl_amd.network import read_xlsx_table
    df = pd.DataFrame({"FROM": [1, 2], "TO": [2, 3], "R": [0.0922, 0.493], "X": [0.047, 0.2511], "Imax": [400, 400]})
    df.to_excel(tmp_path / "t.xlsx", index=False)
    header, rows = read_xlsx_table(str(tmp_path / "t.xlsx"))
    assert header == list(df.columns) and np.allclose(np.array(rows, float), df.to_numpy(float))
