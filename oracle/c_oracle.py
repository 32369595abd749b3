"""TEST INFRASTRUCTURE — ctypes wrapper of oracle/flexenv_oracle.c (the plain-C restatement).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from math import acos, tan

import numpy as np

from .env_oracle import DEFAULT_CFG

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "flexenv_oracle.c")
OUT = os.path.join(HERE, "libflexenv_oracle.so")
OMAXB, OMAXA = 64, 8


def build(force=False):
    """gcc -O3 -march=x86-64-v3 (AVX2/FMA baseline: the .so built in the container also runs on the GPU box's host)."""
    if not force and os.path.exists(OUT) and os.path.getmtime(OUT) >= os.path.getmtime(SRC):
        return OUT
    subprocess.run(["gcc", "-O3", "-march=x86-64-v3", "-fopenmp", "-shared", "-fPIC", "-o", OUT, SRC, "-lm"], check=True)
    return OUT


class OCfg(C.Structure):
    _fields_ = [(k, C.c_int32) for k in ("n_bus", "n_agents", "history", "episode_limit", "raw_actions",
                                         "pf_max_iter", "slack", "n_lines", "pf_solver", "pad0")] + \
               [(k, C.c_double) for k in ("v_min", "v_max", "e_min", "e_max", "p_ch_max", "p_dis_max", "eta_ch",
                                          "eta_dis", "tan_phi", "max_power_reduction", "pv_cost", "ess_cost",
                                          "discomfort_coeff", "voltage_coeff", "dt", "fail_penalty", "pf_tol")] + \
               [("agent_bus", C.c_int32 * OMAXA), ("line_from", C.c_int32 * OMAXB), ("line_to", C.c_int32 * OMAXB),
                ("line_r", C.c_double * OMAXB), ("line_x", C.c_double * OMAXB)]


class OState(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("V", "E", "Einit", "act", "cum", "hist", "steps", "start", "row",
                                          "obscnt", "series")] + [("rows", C.c_int64), ("cols", C.c_int32)]


_lib = None


def load():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        assert _lib.oracle_sizeof_cfg() == C.sizeof(OCfg) and _lib.oracle_sizeof_state() == C.sizeof(OState)
    return _lib


def set_threads(n):
    """Pin the OpenMP team of the batch calls to ``n`` threads; returns the team size actually in force."""
    lib = load()
    lib.oracle_set_threads(int(n))
    return int(lib.oracle_get_threads())


SOLVERS = {"dense_nr": 0, "distflow_sweep": 1}


def make_cfg(net, cfg=None, alg=None, pf_tol=1e-12, pf_max_iter=20, solver="dense_nr"):
    c = dict(DEFAULT_CFG)
    c.update(cfg or {})
    buses = list(net["bus_numbers"])
    idx = {b: i for i, b in enumerate(buses)}
    o = OCfg()
    o.n_bus, o.n_agents, o.history, o.episode_limit = len(buses), len(net["buildings"]), c["history"], c["episode_limit"]
    o.raw_actions = 1 if alg == "safemaddpg" else 0
    o.pf_max_iter = pf_max_iter
    o.pf_solver = SOLVERS[solver]
    o.slack = [i for i, b in enumerate(buses) if net["bus_types"][b] == 1][0]
    lines = list(net["line_connections"])
    o.n_lines = len(lines)
    for k in ("v_min", "v_max", "e_min", "e_max", "p_ch_max", "p_dis_max", "eta_ch", "eta_dis",
              "max_power_reduction", "pv_cost", "ess_cost", "discomfort_coeff", "voltage_coeff"):
        setattr(o, k, float(c[k]))
    o.tan_phi = tan(acos(c["cos_phi_max"]))
    o.dt = 24 / c["episode_limit"]
    o.fail_penalty = 200.0
    o.pf_tol = pf_tol
    for a, b in enumerate(net["buildings"]):
        o.agent_bus[a] = idx[b]
    for l, (f, t) in enumerate(lines):
        o.line_from[l], o.line_to[l] = idx[f], idx[t]
        o.line_r[l], o.line_x[l] = net["line_resistances"][(f, t)], net["line_reactances"][(f, t)]
    return o


def pf_batch(net, pnet, qnet, pf_tol=1e-12, pf_max_iter=20, solver="dense_nr"):
    lib = load()
    o = make_cfg(net, pf_tol=pf_tol, pf_max_iter=pf_max_iter, solver=solver)
    pnet = np.ascontiguousarray(pnet, np.float64)
    qnet = np.ascontiguousarray(qnet, np.float64)
    vm = np.empty_like(pnet)
    iters = np.empty(len(pnet), np.int32)
    lib.oracle_pf_batch(C.byref(o), len(pnet), pnet.ctypes.data_as(C.c_void_p), qnet.ctypes.data_as(C.c_void_p),
                        vm.ctypes.data_as(C.c_void_p), iters.ctypes.data_as(C.c_void_p))
    return vm, iters


class COracleEnv:
    """N scalar environments stepped on the host (OpenMP over envs; OMP_NUM_THREADS picks the cores)."""

    def __init__(self, net, series_table, n, cfg=None, alg=None, solver="dense_nr"):
        self.lib = load()
        self.cfg = make_cfg(net, cfg, alg, solver=solver)
        self.n = n
        nb, na, H = self.cfg.n_bus, self.cfg.n_agents, self.cfg.history
        self.series = np.ascontiguousarray(series_table, np.float64)
        self.V = np.zeros((n, nb)); self.E = np.zeros((n, na)); self.Einit = np.zeros((n, na))
        self.act = np.zeros((n, 4, na)); self.cum = np.zeros(n); self.hist = np.zeros((n, na, H, 6))
        self.steps = np.zeros(n, np.int32); self.start = np.zeros(n, np.int32); self.row = np.zeros(n, np.int32)
        self.obscnt = np.zeros(n, np.int32)
        self.obs = np.zeros((n, na, H * 6), np.float32)
        self.reward = np.zeros(n); self.done = np.zeros(n, np.uint8); self.failed = np.zeros(n, np.uint8)
        self.info = np.zeros((n, 7))
        st = OState()
        for k in ("V", "E", "Einit", "act", "cum", "hist", "steps", "start", "row", "obscnt", "series"):
            setattr(st, k, getattr(self, k).ctypes.data)
        st.rows, st.cols = self.series.shape
        self.st = st

    def reset(self, start, e0, a0):
        start = np.ascontiguousarray(start, np.int32)
        e0 = np.ascontiguousarray(e0, np.float64)
        a0 = np.ascontiguousarray(a0, np.float64)
        self.lib.oracle_env_reset_batch(C.byref(self.cfg), C.byref(self.st), self.n, start.ctypes.data_as(C.c_void_p),
                                        e0.ctypes.data_as(C.c_void_p), a0.ctypes.data_as(C.c_void_p),
                                        self.obs.ctypes.data_as(C.c_void_p), self.failed.ctypes.data_as(C.c_void_p))
        return self.obs

    def step(self, actions):
        a = np.ascontiguousarray(actions, np.float64)
        self.lib.oracle_env_step_batch(C.byref(self.cfg), C.byref(self.st), self.n, a.ctypes.data_as(C.c_void_p),
                                       self.reward.ctypes.data_as(C.c_void_p), self.done.ctypes.data_as(C.c_void_p),
                                       self.info.ctypes.data_as(C.c_void_p), self.failed.ctypes.data_as(C.c_void_p),
                                       self.obs.ctypes.data_as(C.c_void_p))
        return self.reward, self.done, self.info
