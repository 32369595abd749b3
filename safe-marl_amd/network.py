"""Network tables for the power-flow path.

Mirrors ``utils/create_net.py:8-39`` (``create_network``): the result is the
same dict of bus ids, line set, per-unit R/X/Imax, bus types and static P/Q,
keyed by the reference's 1-based bus ids.  On top of that dict this module
derives the flat, index-based tables the HIP kernels consume (``NetTables``):
a rooted orientation of the radial feeder, per-line admittances (the non-zero
Ybus entries) and the leaf->root elimination schedule.

The reference reads ``data/Nodes_33.xlsx`` / ``data/Lines_33.xlsx``; both are
Git-LFS pointers in the reference checkout, so the default
here is the canonical Baran & Wu (1989) IEEE 33-bus radial feeder (SURVEY.md
App. C).  Everything computed with it is "stand-in IEEE-33 (Baran-Wu), not the
reference's xlsx".
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

# Defaults of madrl/args/env_args/flex_provision.yaml:31-33
V_NOM_KV = 12.66
S_NOM_KVA = 1000.0

# FROM TO R[ohm] X[ohm]  (Baran & Wu 1989, 12.66 kV)
IEEE33_LINES: Tuple[Tuple[int, int, float, float], ...] = (
    (1, 2, 0.0922, 0.0470), (2, 3, 0.4930, 0.2511), (3, 4, 0.3660, 0.1864),
    (4, 5, 0.3811, 0.1941), (5, 6, 0.8190, 0.7070), (6, 7, 0.1872, 0.6188),
    (7, 8, 0.7114, 0.2351), (8, 9, 1.0300, 0.7400), (9, 10, 1.0440, 0.7400),
    (10, 11, 0.1966, 0.0650), (11, 12, 0.3744, 0.1238), (12, 13, 1.4680, 1.1550),
    (13, 14, 0.5416, 0.7129), (14, 15, 0.5910, 0.5260), (15, 16, 0.7463, 0.5450),
    (16, 17, 1.2890, 1.7210), (17, 18, 0.7320, 0.5740), (2, 19, 0.1640, 0.1565),
    (19, 20, 1.5042, 1.3554), (20, 21, 0.4095, 0.4784), (21, 22, 0.7089, 0.9373),
    (3, 23, 0.4512, 0.3083), (23, 24, 0.8980, 0.7091), (24, 25, 0.8960, 0.7011),
    (6, 26, 0.2030, 0.1034), (26, 27, 0.2842, 0.1447), (27, 28, 1.0590, 0.9337),
    (28, 29, 0.8042, 0.7006), (29, 30, 0.5075, 0.2585), (30, 31, 0.9744, 0.9630),
    (31, 32, 0.3105, 0.3619), (32, 33, 0.3410, 0.5302),
)

# bus P[kW] Q[kvar]; bus 1 is the slack (Tb == 1) with no load
IEEE33_LOADS: Tuple[Tuple[int, float, float], ...] = (
    (2, 100, 60), (3, 90, 40), (4, 120, 80), (5, 60, 30), (6, 60, 20),
    (7, 200, 100), (8, 200, 100), (9, 60, 20), (10, 60, 20), (11, 45, 30),
    (12, 60, 35), (13, 60, 35), (14, 120, 80), (15, 60, 10), (16, 60, 20),
    (17, 60, 20), (18, 90, 40), (19, 90, 40), (20, 90, 40), (21, 90, 40),
    (22, 90, 40), (23, 90, 50), (24, 420, 200), (25, 420, 200), (26, 60, 25),
    (27, 60, 25), (28, 60, 20), (29, 120, 70), (30, 200, 600), (31, 150, 70),
    (32, 210, 100), (33, 60, 40),
)

# Imax is only used by the out-of-scope OPF (utils/opf.py); a flat ampacity
# keeps the dict complete.
IEEE33_IMAX_A = 400.0

DEFAULT_BUILDINGS = (5, 10, 15, 20, 25)  # flex_provision.yaml:28-30


def ieee33_tables():
    """Rows in the column layout create_net.py:15-24 expects from the xlsx:
    nodes (NODES, Tb, PDn, QDn) and lines (FROM, TO, R, X, Imax)."""
    loads = {b: (p, q) for b, p, q in IEEE33_LOADS}
    nodes = [(b, 1 if b == 1 else 0, *loads.get(b, (0.0, 0.0))) for b in range(1, 34)]
    lines = [(f, t, r, x, IEEE33_IMAX_A) for f, t, r, x in IEEE33_LINES]
    return nodes, lines


def create_network(env_args: Optional[dict] = None, nodes=None, lines=None) -> dict:
    """Same dict as ``create_network()`` (utils/create_net.py:8-39).

    ``nodes`` rows are (NODES, Tb, PDn[kW], QDn[kvar]); ``lines`` rows are
    (FROM, TO, R[ohm], X[ohm], Imax[A]).  Per-unit scaling follows
    create_net.py:17-24: p = PDn / s_nom, r = R / (v_nom**2 * 1000 / s_nom),
    i_max = Imax / (s_nom / v_nom).  With no tables the IEEE-33 stand-in is used.
    """
    env_args = env_args or {}
    v_nom = float(env_args.get("v_nom", V_NOM_KV))
    s_nom = float(env_args.get("s_nom", S_NOM_KVA))
    if nodes is None or lines is None:
        nodes, lines = ieee33_tables()
    z_base = v_nom ** 2 * 1000 / s_nom
    bus_numbers = [int(n[0]) for n in nodes]
    # the reference builds a *set* of tuples and then list()s it (create_net.py:21,29),
    # so its line order is hash order; every consumer indexes by (from, to) so order
    # is immaterial.  Keep file order here.
    line_connections = [(int(l[0]), int(l[1])) for l in lines]
    return {
        "bus_numbers": bus_numbers,
        "line_connections": line_connections,
        "line_resistances": {(int(l[0]), int(l[1])): l[2] / z_base for l in lines},
        "line_reactances": {(int(l[0]), int(l[1])): l[3] / z_base for l in lines},
        "max_line_currents": {(int(l[0]), int(l[1])): l[4] / (s_nom / v_nom) for l in lines},
        "bus_types": {int(n[0]): int(n[1]) for n in nodes},
        "active_power_demand": {int(n[0]): n[2] / s_nom for n in nodes},
        "reactive_power_demand": {int(n[0]): n[3] / s_nom for n in nodes},
        "buildings": list(env_args.get("buildings", DEFAULT_BUILDINGS)),
        "PVs_at_buildings": list(env_args.get("pv_nodes", DEFAULT_BUILDINGS)),
        "ESSs_at_buildings": list(env_args.get("ess_nodes", DEFAULT_BUILDINGS)),
    }


def read_xlsx_table(path: str):
    """First worksheet of an .xlsx workbook as (header, rows) — what ``pd.read_excel(path)`` hands the reference
    (utils/create_net.py:11-12), without openpyxl: an .xlsx file is a zip of XML parts (ECMA-376), and the two tables
    this path reads are plain header + numeric rows.  Shared strings, inline strings and numeric cells are understood;
    empty trailing rows are dropped; a cell missing from a row is None."""
    import re
    import zipfile
    import xml.etree.ElementTree as ET

    ns = {"m": "http://schemas.openxmlformats.org/spreadsheetml/2006/main",
          "r": "http://schemas.openxmlformats.org/officeDocument/2006/relationships",
          "p": "http://schemas.openxmlformats.org/package/2006/relationships"}
    with zipfile.ZipFile(path) as z:
        names = set(z.namelist())
        shared = []
        if "xl/sharedStrings.xml" in names:
            for si in ET.fromstring(z.read("xl/sharedStrings.xml")).findall("m:si", ns):
                shared.append("".join(t.text or "" for t in si.iter("{%s}t" % ns["m"])))
        # first sheet of the workbook, through its relationship (falls back to the conventional part name)
        sheet = "xl/worksheets/sheet1.xml"
        try:
            wb = ET.fromstring(z.read("xl/workbook.xml"))
            first = wb.find("m:sheets", ns).find("m:sheet", ns)
            rid = first.get("{%s}id" % ns["r"])
            rels = ET.fromstring(z.read("xl/_rels/workbook.xml.rels"))
            for rel in rels.findall("p:Relationship", ns):
                if rel.get("Id") == rid:
                    target = rel.get("Target").lstrip("/")
                    sheet = target if target.startswith("xl/") else "xl/" + target
        except (KeyError, AttributeError):
            pass
        root = ET.fromstring(z.read(sheet))

    def col_index(ref):
        letters = re.match(r"[A-Z]+", ref).group(0)
        k = 0
        for ch in letters:
            k = k * 26 + (ord(ch) - 64)
        return k - 1

    table = []
    for row in root.find("m:sheetData", ns).findall("m:row", ns):
        cells = {}
        for pos, c in enumerate(row.findall("m:c", ns)):
            j = col_index(c.get("r")) if c.get("r") else pos
            t = c.get("t")
            v = c.find("m:v", ns)
            if t == "s":
                val = shared[int(v.text)]
            elif t == "inlineStr":
                val = "".join(x.text or "" for x in c.iter("{%s}t" % ns["m"]))
            elif t == "str":
                val = v.text if v is not None else ""
            elif v is None or v.text is None:
                val = None
            else:
                val = float(v.text)
            cells[j] = val
        if cells and any(x is not None for x in cells.values()):
            table.append([cells.get(j) for j in range(max(cells) + 1)])
    if not table:
        raise ValueError(f"{path}: empty worksheet")
    header = [str(h).strip() if h is not None else "" for h in table[0]]
    rows = [r + [None] * (len(header) - len(r)) for r in table[1:]]
    return header, rows


def load_network_xlsx(data_path: str, env_args: Optional[dict] = None) -> dict:
    """``create_network()`` from the reference's own files (utils/create_net.py:11-24): ``Nodes_33.xlsx`` with columns
    NODES, Tb, PDn, QDn and ``Lines_33.xlsx`` with FROM, TO, R, X, Imax — same columns, same per-unit scaling.  The
    workbooks are read with ``read_xlsx_table`` (no openpyxl needed)."""
    def table(name, cols):
        header, rows = read_xlsx_table(f"{data_path}/{name}")
        missing = [c for c in cols if c not in header]
        if missing:
            raise KeyError(f"{name}: missing column(s) {missing} (have {header})")
        idx = [header.index(c) for c in cols]
        return [tuple(r[i] for i in idx) for r in rows if r[idx[0]] is not None]

    nodes = table("Nodes_33.xlsx", ("NODES", "Tb", "PDn", "QDn"))
    lines = table("Lines_33.xlsx", ("FROM", "TO", "R", "X", "Imax"))
    return create_network(env_args, nodes, lines)


MAX_BUS = 64      # one lane per bus in a 64-wide wavefront
MAX_AGENTS = 8
MAX_LEVELS = 64


@dataclass
class NetTables:
    """Index-based view of the network dict for the kernels and the C oracle.

    Bus index = position in ``bus_numbers`` (the order every env array uses,
    env:361-366).  The feeder is rooted at the slack bus; ``parent[i]`` is the
    bus index one step towards the slack (-1 at the slack), and ``r[i], x[i]``
    are the per-unit impedance of the line between ``i`` and ``parent[i]``.
    ``level[i]`` is the distance from the slack; elimination runs from
    ``n_levels-1`` down to 1 and back-substitution from 1 up.
    """
    n_bus: int
    slack: int
    parent: np.ndarray        # int32 [n_bus]
    level: np.ndarray         # int32 [n_bus]
    r: np.ndarray             # float64 [n_bus] (0 at slack)
    x: np.ndarray             # float64 [n_bus]
    n_levels: int
    child: np.ndarray         # int32 [n_bus, max_children] (-1 padded)
    max_children: int
    line_of_bus: List[Optional[Tuple[int, int]]]  # reference (from,to) key per non-slack bus
    line_forward: np.ndarray  # bool [n_bus]: True if the reference key is (parent, bus)
    agent_bus: np.ndarray     # int32 [n_agents]: bus index of each building

    @property
    def g(self):
        d = self.r ** 2 + self.x ** 2
        return np.where(d > 0, self.r / np.where(d > 0, d, 1.0), 0.0)

    @property
    def b(self):
        d = self.r ** 2 + self.x ** 2
        return np.where(d > 0, -self.x / np.where(d > 0, d, 1.0), 0.0)


def build_tables(net: dict) -> NetTables:
    """Orient the radial feeder away from the slack and flatten it.

    pf.py fixes Vsqr=1 and frees Ps/Qs exactly where ``bus_types == 1``
    (pf.py:51-56); one such bus is required.  The DistFlow model of
    pf.py:65-94 is square only for a tree, so anything else
    is rejected here rather than silently mis-solved.
    """
    buses = list(net["bus_numbers"])
    n = len(buses)
    if n > MAX_BUS:
        raise ValueError(f"{n} buses: the one-wavefront-per-env kernels hold one bus per lane (max {MAX_BUS})")
    idx = {b: i for i, b in enumerate(buses)}
    slacks = [b for b in buses if net["bus_types"][b] == 1]
    if len(slacks) != 1:
        raise ValueError(f"exactly one slack bus (bus_types == 1) is required, got {slacks}")
    lines = list(net["line_connections"])
    if len(lines) != n - 1:
        raise ValueError(f"radial feeder required: {n} buses need {n - 1} lines, got {len(lines)}")
    adj: Dict[int, List[Tuple[int, Tuple[int, int]]]] = {i: [] for i in range(n)}
    for (f, t) in lines:
        adj[idx[f]].append((idx[t], (f, t)))
        adj[idx[t]].append((idx[f], (f, t)))
    parent = np.full(n, -1, np.int32)
    level = np.full(n, -1, np.int32)
    r = np.zeros(n)
    x = np.zeros(n)
    line_of_bus: List[Optional[Tuple[int, int]]] = [None] * n
    line_forward = np.zeros(n, bool)
    root = idx[slacks[0]]
    level[root] = 0
    frontier = [root]
    while frontier:
        nxt = []
        for u in frontier:
            for v, key in adj[u]:
                if level[v] >= 0:
                    continue
                level[v] = level[u] + 1
                parent[v] = u
                r[v] = net["line_resistances"][key]
                x[v] = net["line_reactances"][key]
                line_of_bus[v] = key
                line_forward[v] = (idx[key[1]] == v)
                nxt.append(v)
        frontier = nxt
    if (level < 0).any():
        raise ValueError("network is not connected to the slack bus")
    kids: List[List[int]] = [[] for _ in range(n)]
    for i in range(n):
        if parent[i] >= 0:
            kids[parent[i]].append(i)
    mc = max(1, max(len(k) for k in kids))
    child = np.full((n, mc), -1, np.int32)
    for i, k in enumerate(kids):
        child[i, :len(k)] = k
    agents = list(net["buildings"])
    if len(agents) > MAX_AGENTS:
        raise ValueError(f"{len(agents)} buildings: max {MAX_AGENTS}")
    # env:379,382 index current_pv_power / current_ess_energy by the building id
    if list(net["PVs_at_buildings"]) != agents or list(net["ESSs_at_buildings"]) != agents:
        raise ValueError("pv_nodes and ess_nodes must equal buildings (env:379-382, SURVEY A15)")
    return NetTables(
        n_bus=n, slack=root, parent=parent, level=level, r=r, x=x,
        n_levels=int(level.max()) + 1, child=child, max_children=mc,
        line_of_bus=line_of_bus, line_forward=line_forward,
        agent_bus=np.array([idx[b] for b in agents], np.int32),
    )
