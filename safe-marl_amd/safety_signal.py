"""Voltage predictor behind the safety layer.

The reference fits it offline — safety_signal/data_generation.py:14-58 draws 1000 scenarios at +-30 % of
the static bus loads and solves each with IPOPT, train_safety_signal_model.py:30-46,73 MinMax-scales
inputs and outputs, splits 80/20 with seed 42 and fits one ordinary-least-squares model per bus — and
ships the result as ``linear_multioutput_regressor.pkl`` (safemaddpg.py:27), which is not in the
repository.  Here the scenarios are solved in one batched HIP power-flow launch and the fit is plain
NumPy least squares; ``coef_`` / ``intercept_`` have the shapes sklearn's estimators expose, and
``building_terms`` applies the consumer's own slicing (safemaddpg.py:182-184, 266: quirks A11/A12).
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np


def draw_scenarios(net, num_scenarios=1000, variation=0.3, rng=None):
    """data_generation.py:31-36: per scenario, P then Q, one uniform per bus in bus order.
    Returns (P [num, n_bus], Q [num, n_bus])."""
    rng = np.random if rng is None else rng
    buses = list(net["bus_numbers"])
    pb = np.array([net["active_power_demand"][b] for b in buses])
    qb = np.array([net["reactive_power_demand"][b] for b in buses])
    P = np.empty((num_scenarios, len(buses)))
    Q = np.empty_like(P)
    for s in range(num_scenarios):
        P[s] = pb * (1 + rng.uniform(-variation, variation, len(buses)))
        Q[s] = qb * (1 + rng.uniform(-variation, variation, len(buses)))
    return P, Q


def interleave(P, Q):
    """data_generation.py:48-49: feature columns [P_1, Q_1, ..., P_n, Q_n]."""
    X = np.empty((P.shape[0], 2 * P.shape[1]))
    X[:, 0::2] = P
    X[:, 1::2] = Q
    return X


def _minmax(a):
    """sklearn MinMaxScaler(feature_range=(0,1)): constant columns keep scale 1 (-> all zeros)."""
    lo, hi = a.min(0), a.max(0)
    rng = hi - lo
    rng[rng == 0.0] = 1.0
    return (a - lo) / rng


def _split_80_20(n, seed=42):
    """sklearn train_test_split(test_size=0.2, random_state=42): ShuffleSplit's permutation, test first."""
    perm = np.random.RandomState(seed).permutation(n)
    n_test = int(np.ceil(0.2 * n))
    n_train = int(np.floor(0.8 * n))
    return perm[n_test:n_test + n_train], perm[:n_test]


@dataclass
class VoltagePredictor:
    coef_: np.ndarray        # [n_bus, 2*n_bus]   one row per output estimator (est.coef_)
    intercept_: np.ndarray   # [n_bus]
    test_mse: float = float("nan")

    def predict(self, X):
        return X @ self.coef_.T + self.intercept_

    def consumer_split(self):
        """safemaddpg.py:182-184: W_P = coef[:, :n_bus], W_Q = coef[:, n_bus:], b = intercept (A12)."""
        n = self.intercept_.shape[0]
        return self.coef_[:, :n], self.coef_[:, n:], self.intercept_

    def building_terms(self, net):
        """Row sums used by the own-bus constraint (safemaddpg.py:266,272): per building (s_p, s_q, beta)."""
        W_P, W_Q, b = self.consumer_split()
        idx = [list(net["bus_numbers"]).index(bus) for bus in net["buildings"]]
        return W_P[idx].sum(1), W_Q[idx].sum(1), b[idx]


def fit_from_data(X, Y, seed=42):
    """train_safety_signal_model.py:34-46,73: MinMax-scale X and Y, 80/20 split (seed 42), OLS with
    intercept per output (LinearRegression = centred least squares, minimum-norm on rank deficiency)."""
    Xs, Ys = _minmax(np.asarray(X, float)), _minmax(np.asarray(Y, float))
    tr, te = _split_80_20(len(Xs), seed)
    xm, ym = Xs[tr].mean(0), Ys[tr].mean(0)
    coef, *_ = np.linalg.lstsq(Xs[tr] - xm, Ys[tr] - ym, rcond=None)      # [2n, n]
    coef = coef.T
    intercept = ym - coef @ xm
    mse = float(np.mean((Xs[te] @ coef.T + intercept - Ys[te]) ** 2))
    return VoltagePredictor(coef, intercept, mse)


def fit_voltage_predictor(net, device="cuda:0", num_scenarios=1000, variation=0.3, seed=0):
    """The whole offline pipeline: scenarios -> batched HIP power flow -> fit."""
    import torch
    from .flex_env import pf_solve_batch
    P, Q = draw_scenarios(net, num_scenarios, variation, np.random.RandomState(seed))
    out = pf_solve_batch(net, torch.from_numpy(P).to(device), torch.from_numpy(Q).to(device))
    ok = ~out["failed"].bool().cpu().numpy()            # data_generation.py:56-58 skips failed scenarios
    V = out["v"].cpu().numpy()
    return fit_from_data(interleave(P[ok], Q[ok]), V[ok])
