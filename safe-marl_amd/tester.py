"""Evaluation rollouts with the reference's record format (utils/tester.py:6-70).

``PGTester(args, behaviour_net, env).run(day, hour, quarter)`` returns the dict ``test_agent.py:84-96`` pickles:
keys pv_active, pv_reactive, bus_active, bus_reactive, bus_voltage, ess_energy, power_reduction, ess_charging,
ess_discharging, price — one entry after the manual reset and one per step.  With the N=1 drop-in env the
entries are the 1-D arrays the reference stores; with a ``VecFlexProvisionEnv`` every entry gains a leading
environment axis and ``day/hour/quarter`` may be arrays (one start time per environment)."""
from __future__ import annotations

import numpy as np
import torch as th

from .util import prep_obs, translate_action

RECORD_KEYS = ("pv_active", "pv_reactive", "bus_active", "bus_reactive", "bus_voltage", "ess_energy",
               "power_reduction", "ess_charging", "ess_discharging", "price")


class PGTester(object):
    def __init__(self, args, behaviour_net, env):
        self.args, self.env = args, env
        self.device = th.device("cuda" if th.cuda.is_available() and args.cuda else "cpu")
        self.behaviour_net = behaviour_net.to(self.device).eval()
        self.n_, self.obs_dim, self.act_dim = args.agent_num, args.obs_size, args.action_dim

    # -- N = 1, the reference's loop (tester.py:16-70) --------------------------------------------------------
    def _snapshot_single(self, record):
        e = self.env
        for key, fn in zip(RECORD_KEYS, (e._get_pv_active, e._get_pv_reactive, e._get_bus_active, e._get_bus_reactive,
                                         e._get_bus_v, e._get_ess_energy, e._get_power_reduction, e._get_ess_charging,
                                         e._get_ess_discharging, e._get_price)):
            record[key].append(fn())

    def _run_single(self, day, hour, quarter):
        state, _ = self.env.manual_reset(day, hour, quarter)
        last_hid = self.behaviour_net.policy_dicts[0].init_hidden()
        record = {k: [] for k in RECORD_KEYS}
        self._snapshot_single(record)
        avail = th.tensor(self.env.get_avail_actions())
        for t in range(self.args.max_steps):
            state_ = prep_obs(state).contiguous().view(1, self.n_, self.obs_dim).to(self.device)
            with th.no_grad():
                action, _, _, _, hid = self.behaviour_net.get_actions(state_, status="test", exploration=False,
                                                                      actions_avail=avail, target=False, last_hid=last_hid)
            _, actual = translate_action(self.args, action, self.env)
            _, done, _ = self.env.step(actual)
            self._snapshot_single(record)
            state, last_hid = self.env.get_obs(), hid
            if done or t == self.args.max_steps - 1:
                break
        return record

    # -- N envs at once -------------------------------------------------------------------------------------------
    def _snapshot_vec(self, record):
        v = self.env
        nb, na = v.n_bus, v.n_agents
        rows = v.series_dev.index_select(0, v.peek("ROW").long())                     # current data row per env
        record["bus_active"].append(rows[:, :nb].cpu().numpy())
        record["bus_reactive"].append(rows[:, nb:2 * nb].cpu().numpy())
        record["pv_active"].append(rows[:, 2 * nb:2 * nb + na].cpu().numpy())
        record["price"].append(rows[:, 2 * nb + na:].cpu().numpy())
        for key, field in (("pv_reactive", "QPV"), ("bus_voltage", "V"), ("ess_energy", "E"),
                           ("power_reduction", "PRED"), ("ess_charging", "CH"), ("ess_discharging", "DIS")):
            record[key].append(v.peek(field).cpu().numpy())

    def _run_vec(self, day, hour, quarter, e0=None, a0=None):
        v = self.env
        N = v.n_envs
        spec = dict(day=np.broadcast_to(np.asarray(day, np.int32), (N,)).copy(),
                    hour=np.broadcast_to(np.asarray(hour, np.int32), (N,)).copy(),
                    interval=np.broadcast_to(np.asarray(quarter, np.int32), (N,)).copy(), e0=e0, a0=a0)
        obs = v.reset(spec=spec).clone()
        if int(v.failed.sum().item()):
            raise RuntimeError("The power flow for the current initialization cannot be solved.")
        last_hid = th.zeros(N, self.n_, self.args.hid_size, device=self.device)
        avail = th.ones(N, self.n_, self.act_dim, device=self.device)
        record = {k: [] for k in RECORD_KEYS}
        record["actions"] = []                 # extra: what the env received, so a record can be re-simulated
        self._snapshot_vec(record)
        horizon = min(self.args.max_steps, v.episode_limit - 1)
        for t in range(horizon):
            with th.no_grad():
                action, _, _, _, hid = self.behaviour_net.get_actions(obs, status="test", exploration=False,
                                                                      actions_avail=avail, target=False, last_hid=last_hid)
                actual = self.behaviour_net.env_action(action)
            v.step(actual, fuse_obs=True)
            record["actions"].append(actual.cpu().numpy())
            self._snapshot_vec(record)
            obs, last_hid = v.obs.clone(), hid
        return record

    def run(self, day, hour, quarter, **spec):
        if hasattr(self.env, "handle"):
            return self._run_vec(day, hour, quarter, **spec)
        return self._run_single(day, hour, quarter)
