"""Time-series table for the environment: one fp64 row per 15-min interval,
columns ``[Pd(n_bus) | Qd(n_bus) | Ppv(n_agents) | price(1)]`` with the slack
column of Pd/Qd held at zero (env:489-490, 510-511).  Row-major so that one
wavefront reads its env's current row with two coalesced loads.

The reference builds these from data/load_active.csv, load_reactive.csv,
pv_active.csv and prices.csv (env:431-471); all four are Git-LFS pointers in the
reference checkout, so the default is the synthetic generator of SURVEY.md §8(d).
``from_frames`` ingests real data in the reference's on-disk format.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np


@dataclass
class SeriesTable:
    table: np.ndarray        # float64 [rows, 2*n_bus + n_agents + 1]
    n_bus: int
    n_agents: int
    time_delta: int = 15     # minutes per row (env:422)

    @property
    def rows(self):
        return self.table.shape[0]

    @property
    def cols(self):
        return self.table.shape[1]

    @property
    def per_hour(self):
        return 60 // self.time_delta

    @property
    def pv_days(self):
        """(index[-1] - index[0]).days of env:421."""
        return (self.rows - 1) * self.time_delta // (24 * 60)

    def n_start_days(self, episode_limit):
        episode_days = (episode_limit // (24 * self.per_hour)) + 1   # env:423
        return self.pv_days - episode_days                           # env:424

    # views in the shapes the reference's episode slices have
    @property
    def active(self):
        return self.table[:, :self.n_bus]

    @property
    def reactive(self):
        return self.table[:, self.n_bus:2 * self.n_bus]

    @property
    def pv(self):
        return self.table[:, 2 * self.n_bus:2 * self.n_bus + self.n_agents]

    @property
    def price(self):
        return self.table[:, -1]


def make_synthetic_series(net, n_days=1096, seed=20250114, pv_scale=0.15, time_delta=15) -> SeriesTable:
    """Pd = P_base*(0.6+0.4 sin^2(pi h/24))*U(0.8,1.2), Qd = Pd*Q_base/P_base,
    Ppv = pv_scale*max(0, sin(pi (h-6)/12))*U(0.7,1.0), price = U(0.05,0.30)."""
    buses = list(net["bus_numbers"])
    n_bus = len(buses)
    agents = list(net["buildings"])
    per_day = 24 * 60 // time_delta
    rows = n_days * per_day
    rng = np.random.default_rng(seed)
    h = (np.arange(rows) % per_day) * (time_delta / 60.0)
    pb = np.array([net["active_power_demand"][b] for b in buses])
    qb = np.array([net["reactive_power_demand"][b] for b in buses])
    ratio = np.divide(qb, pb, out=np.zeros_like(qb), where=pb > 0)
    shape = (0.6 + 0.4 * np.sin(np.pi * h / 24) ** 2)[:, None]
    pd = pb[None, :] * shape * rng.uniform(0.8, 1.2, (rows, n_bus))
    slack = [i for i, b in enumerate(buses) if net["bus_types"][b] == 1]
    pd[:, slack] = 0.0
    qd = pd * ratio[None, :]
    pv = pv_scale * np.maximum(0.0, np.sin(np.pi * (h - 6) / 12))[:, None] * rng.uniform(0.7, 1.0, (rows, len(agents)))
    price = rng.uniform(0.05, 0.30, (rows, 1))
    return SeriesTable(np.ascontiguousarray(np.hstack([pd, qd, pv, price])), n_bus, len(agents), time_delta)


def from_frames(net, active, reactive, pv, price, time_delta=15) -> SeriesTable:
    """Already-resampled arrays in the reference's column layout: ``active``/``reactive``
    [rows, n_bus-1] (no slack column, env:488-490), ``pv`` [rows, n_agents], ``price`` [rows]
    or [rows, 1].  The slack column is inserted as zeros in front, as env:489-490 does."""
    active, reactive, pv = (np.asarray(a, float) for a in (active, reactive, pv))
    price = np.asarray(price, float).reshape(len(active), -1)
    if price.shape[1] != 1:
        raise ValueError("price must be a single column (env:681,689; SURVEY A15)")
    z = np.zeros((len(active), 1))
    n_bus = len(net["bus_numbers"])
    if active.shape[1] != n_bus - 1 or reactive.shape[1] != n_bus - 1:
        raise ValueError("load tables need one column per non-slack bus")
    return SeriesTable(np.ascontiguousarray(np.hstack([z, active, z, reactive, pv, price])),
                       n_bus, pv.shape[1], time_delta)


def load_csv_dir(net, data_path, env_args=None) -> SeriesTable:
    """The reference's on-disk format (env:431-471): ``load_active.csv``, ``load_reactive.csv``,
    ``pv_active.csv``, ``prices.csv``, each with a time column followed by one column per non-slack bus /
    per PV / the price.  Scaled by demand_scale / reactive_scale / pv_scale (env:437,446,455), resampled to
    ``sample_interval`` by mean and linearly interpolated (env:467-471) — with pandas, exactly as the reference does.
    """
    import os
    import pandas as pd
    a = dict(pv_scale=0.15, demand_scale=1.0, reactive_scale=1.0, sample_interval="15min")
    a.update(env_args or {})

    def load(name, scale):
        df = pd.read_csv(os.path.join(data_path, name), index_col=None)
        df.index = pd.to_datetime(df.iloc[:, 0])
        df.index.name = "time"
        df = df.iloc[::1, 1:] * scale
        return df.resample(a["sample_interval"]).mean().interpolate(method="linear")

    active = load("load_active.csv", a["demand_scale"])
    reactive = load("load_reactive.csv", a["reactive_scale"])
    pv = load("pv_active.csv", a["pv_scale"])
    price = load("prices.csv", 1.0)
    n = min(len(active), len(reactive), len(pv), len(price))
    time_delta = int((pv.index[1] - pv.index[0]).seconds // 60)             # env:422
    return from_frames(net, active.values[:n], reactive.values[:n], pv.values[:n], price.values[:n], time_delta)
