"""TEST INFRASTRUCTURE — scalar CPU restatement of the flexibility-provision
environment.  Not product code: only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import it.

PINNED (round 4) to the reference's EXECUTED environment code for everything but the
numerical power-flow solve: tests/golden/make_env_golden.py imports and runs
flexibility_provision_env.py and create_net.py from /root/reference — with the
solve rebound to oracle/pf_oracle.py (pyomo / IPOPT are absent) and stand-in
workbooks / CSVs (the reference's are LFS pointers) — and tests/test_env_golden_cpu.py
holds this file to those runs at 1e-12: whole episodes of reset / step / get_obs /
get_state / reward terms, a second reset, manual_reset, the raw-action branch, an
injected solver failure, the global-NumPy draw order.  PARITY UNPINNED for the solve
itself against IPOPT (oracle/pf_oracle.py, oracle/pf_nlp_oracle.py).  This file
follows the reference line by line; every block cites what it restates (paths under
/root/reference/madrl/environments/flex_provision/flexibility_provision_env.py
unless noted).

Deliberately bug-compatible: A2 data-row lag, A4 ESS clip
without dt, A5 initial_ess_energy not refreshed by reset, A6 signed der_cost,
A7 penalty over all buses, A8 reward from restored actions on failure,
A9 cumulative_reward excludes the current step, A16 zero left-padding.
"""
from __future__ import annotations

from math import acos, tan

import numpy as np

from . import pf_oracle

DEFAULT_CFG = dict(  # madrl/args/env_args/flex_provision.yaml:3-33
    history=24, v_max=1.1, v_min=0.9, episode_limit=96, action_low=0, action_high=1.0, seed=0,
    e_min=0.0, e_max=0.025, pv_cost=0.05, ess_cost=0.03, discomfort_coeff=0.15, voltage_coeff=1.0,
    p_ch_max=0.005, p_dis_max=0.005, eta_ch=0.9, eta_dis=0.9, cos_phi_max=0.95,
    max_power_reduction=0.5,
)

# ---------------------------------------------------------------------------
# Reset stream of the *vectorised* env (the build's own spec, DESIGN.md
# "reset stream"): Philox4x32-10, key = 64-bit seed, counter =
# (block j, episode counter c, env id e, TAG).  Restated here so the HIP reset
# kernel can be checked draw for draw.
# ---------------------------------------------------------------------------
PHILOX_M0 = 0xD2511F53
PHILOX_M1 = 0xCD9E8D57
PHILOX_W0 = 0x9E3779B9
PHILOX_W1 = 0xBB67AE85
RESET_TAG = 0x5AFE0001


def philox4x32_10(counter, key):
    c0, c1, c2, c3 = (int(c) & 0xFFFFFFFF for c in counter)
    k0, k1 = (int(k) & 0xFFFFFFFF for k in key)
    for _ in range(10):
        p0 = PHILOX_M0 * c0
        p1 = PHILOX_M1 * c2
        hi0, lo0 = p0 >> 32, p0 & 0xFFFFFFFF
        hi1, lo1 = p1 >> 32, p1 & 0xFFFFFFFF
        c0, c1, c2, c3 = (hi1 ^ c1 ^ k0) & 0xFFFFFFFF, lo1, (hi0 ^ c3 ^ k1) & 0xFFFFFFFF, lo0
        k0 = (k0 + PHILOX_W0) & 0xFFFFFFFF
        k1 = (k1 + PHILOX_W1) & 0xFFFFFFFF
    return c0, c1, c2, c3


def philox_uniform_pair(block, episode, env, seed):
    """Two doubles in [0,1) with 53 random bits each."""
    x = philox4x32_10((block, episode, env, RESET_TAG), (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF))
    u0 = (((x[0] << 32) | x[1]) >> 11) * (1.0 / 9007199254740992.0)
    u1 = (((x[2] << 32) | x[3]) >> 11) * (1.0 / 9007199254740992.0)
    return u0, u1


def reset_draws(env, episode, seed, n_agents, n_start_days, per_hour, cfg):
    """(day, hour, interval, E0[n_agents], a0[4*n_agents]) for one reset attempt; the draw
    order is the reference's: hour, day, interval (env:85-87), E0 (env:100), a0 (env:103)."""
    ndraw = 3 + n_agents + 4 * n_agents
    u = []
    for j in range((ndraw + 1) // 2):
        u.extend(philox_uniform_pair(j, episode, env, seed))
    hour = int(u[0] * 24)
    day = int(u[1] * n_start_days)
    interval = int(u[2] * per_hour)
    lo, hi = 0.9 * (cfg["e_max"] / 2), 1.1 * (cfg["e_max"] / 2)
    e0 = [lo + (hi - lo) * u[3 + k] for k in range(n_agents)]
    al, ah = cfg["action_low"], cfg["action_high"]
    a0 = [al + (ah - al) * u[3 + n_agents + k] for k in range(4 * n_agents)]
    return day, hour, interval, e0, a0


class FlexEnvOracle:
    """One environment, reference semantics.

    ``active``/``reactive`` are [rows, n_bus] with the slack column already prepended
    as zeros (env:489-490, 510-511); ``pv`` is [rows, n_agents] (already scaled by
    pv_scale, env:437); ``price`` is [rows].  ``time_delta`` is minutes per row
    (env:422) and ``pv_days`` the day count used by ``_select_start_day`` (env:421).
    """

    def __init__(self, net, cfg, active, reactive, pv, price, time_delta=15, pv_days=None, alg=None):
        self.net = net
        self.cfg = dict(DEFAULT_CFG)
        self.cfg.update(cfg or {})
        self.alg = alg                                   # env:43
        self.active, self.reactive, self.pv = np.asarray(active), np.asarray(reactive), np.asarray(pv)
        self.price = np.asarray(price).reshape(-1)
        self.time_delta = int(time_delta)
        self.per_hour = 60 // self.time_delta
        rows = self.active.shape[0]
        self.pv_days = int(pv_days) if pv_days is not None else (rows - 1) * self.time_delta // (24 * 60)
        self.buses = list(net["bus_numbers"])
        self.bidx = {b: i for i, b in enumerate(self.buses)}
        self.buildings = list(net["buildings"])
        self.n_agents = len(self.buildings)              # env:66
        self.n_actions = 4                               # env:67
        self.history = self.cfg["history"]
        self.episode_limit = self.cfg["episode_limit"]
        self.tan_phi = tan(acos(self.cfg["cos_phi_max"]))  # env:623

    # -- episode sampling (env:410-424, 473-478) ---------------------------------
    def n_start_days(self):
        episode_days = (self.episode_limit // (24 * self.per_hour)) + 1          # env:423
        return self.pv_days - episode_days                                       # env:424

    def _start_row(self, day, hour, interval):
        return interval + hour * self.per_hour + day * 24 * self.per_hour          # env:477

    # -- helpers (env:621-677) ----------------------------------------------------
    def _scale_and_clip_q_pv(self, a, ppv):                                        # env:621-626
        c = self.tan_phi * ppv
        return float(np.clip(-c + a * (c - (-c)), -c, c))

    def _clip_power_charging_discharging(self, ch, dis, e):                        # env:628-661
        c = self.cfg
        ch = float(np.clip(ch, 0, c["p_ch_max"]))
        dis = float(np.clip(dis, 0, c["p_dis_max"]))
        e_next = e + c["eta_ch"] * ch - (1 / c["eta_dis"]) * dis                   # env:634 (no dt: A4)
        if e_next > c["e_max"]:
            excess = e_next - c["e_max"]
            if ch > excess / c["eta_ch"]:
                ch -= excess / c["eta_ch"]
            else:
                dis += (excess - ch * c["eta_ch"]) * c["eta_dis"]
                ch = 0
        elif e_next < c["e_min"]:
            lack = c["e_min"] - e_next
            if dis > lack * c["eta_dis"]:
                dis -= lack * c["eta_dis"]
            else:
                ch += (lack - dis / c["eta_dis"]) / c["eta_ch"]
                dis = 0
        ch = float(np.clip(ch, 0, c["p_ch_max"]))                                  # env:658-659
        dis = float(np.clip(dis, 0, c["p_dis_max"]))
        return ch, dis

    @staticmethod
    def _adjust_ess(ch, dis):                                                      # env:663-674
        if ch > 0 and dis > 0:
            if ch > dis:
                return ch - dis, 0.0
            return 0.0, dis - ch
        return ch, dis

    def _load_row(self, t):                                                        # env:609-619
        r = self.start + t
        self.cur_pd = self.active[r].astype(float)
        self.cur_qd = self.reactive[r].astype(float)
        self.cur_pv = self.pv[r].astype(float)
        self.cur_price = float(self.price[r])

    def _parse(self, actions, e_for_clip, scaled):
        """env:262-293 (step) / env:113-130 (reset).  Returns per-agent lists."""
        c = self.cfg
        pr, ch, dis, q = [], [], [], []
        for i, b in enumerate(self.buildings):
            if scaled:                                                             # env:276-281
                p = c["max_power_reduction"] * actions[i * 4]
                cc = c["p_ch_max"] * actions[i * 4 + 1]
                dd = c["p_dis_max"] * actions[i * 4 + 2]
                qq = self._scale_and_clip_q_pv(actions[i * 4 + 3], self.cur_pv[i])
            else:                                                                  # env:268-274 (safemaddpg)
                p, cc, dd, qq = (float(actions[i * 4 + k]) for k in range(4))
            p = float(np.clip(p, 0, c["max_power_reduction"]))                     # env:284, 677
            cc, dd = self._adjust_ess(cc, dd)                                      # env:287
            cc, dd = self._clip_power_charging_discharging(cc, dd, e_for_clip[i])  # env:289-290
            pr.append(p); ch.append(cc); dis.append(dd); q.append(qq)
        pred = [self.cur_pd[self.bidx[b]] * pr[i] for i, b in enumerate(self.buildings)]  # env:293
        return pr, pred, ch, dis, q

    def _solve(self, pred, ch, dis, q, e_init):
        """pf.py:10-113 via the oracle; returns (V[n_bus], E_next[n_agents])."""
        pnet = self.cur_pd.copy()
        qnet = self.cur_qd.copy()
        for i, b in enumerate(self.buildings):
            k = self.bidx[b]
            pnet[k] += -pred[i] - self.cur_pv[i] + ch[i] - dis[i]                  # pf.py:69-73
            qnet[k] -= q[i]                                                        # pf.py:81-82
        vm, _, _ = pf_oracle.nr_polar(self.net, pnet, qnet)
        dt = 24 / self.episode_limit                                               # pf.py:23-24
        e_next = [e_init[i] + dt * (self.cfg["eta_ch"] * ch[i] - (1 / self.cfg["eta_dis"]) * dis[i])
                  for i in range(self.n_agents)]                                   # pf.py:96-98
        # pf.py:41-45: E_next is a NonNegativeReals variable pinned by the equality above: negative -> infeasible NLP ->
        # pf.py:104-105 raises (pf_oracle.DOMAIN_EPS: IPOPT's default bound relaxation)
        if any(not (e >= -pf_oracle.DOMAIN_EPS) for e in e_next) or not np.all(np.isfinite(vm)):
            raise pf_oracle.SolverFailed("variable outside its declared domain (pf.py:41-45)")
        return vm, e_next

    # -- reset (env:74-155 / 157-239) ---------------------------------------------
    def reset(self, spec=None):
        """``spec`` = (day, hour, interval, E0[n_agents], a0[4 n_agents]) injects the draws;
        None draws them from the global NumPy RNG in the reference's order (env:85-87,100,103)."""
        self.steps = 1                                                             # env:76
        self.cumulative_reward = 0.0
        self.obs_history = [[] for _ in range(self.n_agents)]
        while True:
            if spec is None:
                hour = np.random.choice(24)                                        # env:412
                day = np.random.choice(self.n_start_days())                        # env:424
                interval = np.random.choice(self.per_hour)                         # env:416
            else:
                day, hour, interval = spec[0], spec[1], spec[2]
            self.start = self._start_row(day, hour, interval)
            self._load_row(self.steps)                                             # env:98
            c = self.cfg
            if spec is None:
                e0 = [np.random.uniform(0.9 * (c["e_max"] / 2), 1.1 * (c["e_max"] / 2))
                      for _ in range(self.n_agents)]                               # env:100
                a0 = np.random.uniform(low=c["action_low"], high=c["action_high"],
                                       size=self.n_agents * self.n_actions)        # env:716-719
            else:
                e0, a0 = list(spec[3]), np.asarray(spec[4], float)
            self.initial_ess_energy = list(e0)
            _, self.power_reduction, self.ess_charging, self.ess_discharging, self.q_pv = \
                self._parse(a0, self.initial_ess_energy, scaled=True)              # env:113-130 (always scaled)
            try:
                self.current_voltage, self.current_ess_energy = self._solve(
                    self.power_reduction, self.ess_charging, self.ess_discharging, self.q_pv,
                    self.initial_ess_energy)                                       # env:134-147
                break
            except pf_oracle.SolverFailed:
                if spec is not None:
                    raise
        # A5: initial_ess_energy stays the pre-solve draw.
        return self.get_obs(), self.get_state()

    # -- step (env:241-356) -----------------------------------------------------------
    def step(self, actions):
        actions = np.asarray(actions).reshape(self.n_agents * self.n_actions)      # env:260
        last = (self.power_reduction, self.ess_charging, self.ess_discharging, self.q_pv)
        _, pred, ch, dis, q = self._parse(actions, self.current_ess_energy,
                                          scaled=(self.alg != "safemaddpg"))       # env:268-293
        solvable = False
        try:
            v, e = self._solve(pred, ch, dis, q, self.initial_ess_energy)          # env:298-308 (E_init!)
            self.current_voltage, self.current_ess_energy = v, e
            self.power_reduction, self.ess_charging, self.ess_discharging, self.q_pv = pred, ch, dis, q
            solvable = True
        except pf_oracle.SolverFailed:
            self.power_reduction, self.ess_charging, self.ess_discharging, self.q_pv = last  # env:319-328
        reward, info = self.calculate_reward()                                     # env:330-335
        if not solvable:
            reward -= 200                                                          # env:336
            info["solver_failed"] = True                                           # env:337
        self._load_row(self.steps)                                                 # env:340 (A2)
        self.steps += 1                                                            # env:342
        self.cumulative_reward += reward
        terminated = bool(self.steps >= self.episode_limit or not solvable)        # env:345
        self.initial_ess_energy = list(self.current_ess_energy)                    # env:354
        return reward, terminated, info

    def calculate_reward(self):                                                    # env:679-706
        c = self.cfg
        revenue = sum(self.cur_price * p for p in self.power_reduction)
        der_cost = sum(c["pv_cost"] * q for q in self.q_pv)
        ess_cost = sum(c["ess_cost"] * (a + b) for a, b in zip(self.ess_charging, self.ess_discharging))
        discomfort = sum(c["discomfort_coeff"] * p ** 2 for p in self.power_reduction)
        vpen = sum(c["voltage_coeff"] * max(0, v - c["v_max"], c["v_min"] - v) for v in self.current_voltage)
        reward = revenue - der_cost - ess_cost - discomfort - vpen
        info = {"reward": float(reward), "revenue": float(revenue), "der_cost": float(der_cost),
                "ess_cost": float(ess_cost), "discomfort_penalty": float(discomfort),
                "voltage_penalty": float(vpen), "cumulative_reward": self.cumulative_reward}
        return float(reward), info

    # -- observations (env:358-403) ---------------------------------------------------
    def get_state(self):
        return np.concatenate([self.cur_pd, self.cur_qd, self.cur_pv, self.current_voltage,
                               [self.cur_price], self.current_ess_energy])

    def get_obs(self):
        out = []
        for i, b in enumerate(self.buildings):
            k = self.bidx[b]
            obs = np.array([self.cur_pd[k], self.cur_qd[k], self.cur_pv[i], self.current_voltage[k],
                            self.cur_price, self.current_ess_energy[i]])
            hist = self.obs_history[i]
            if self.history > 1:
                if len(hist) >= self.history - 1:
                    full = np.concatenate(hist[-self.history + 1:] + [obs])
                else:
                    zeros = [np.zeros_like(obs)] * (self.history - len(hist) - 1)
                    full = np.concatenate(zeros + hist + [obs])
                hist.append(obs.copy())
                out.append(full)
            else:
                out.append(obs)
        return out
