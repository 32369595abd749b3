import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import safe_marl_amd  # noqa: E402,F401  (import shim for the hyphenated package dir)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def net():
    from safe_marl_amd.network import create_network
    return create_network()


@pytest.fixture(scope="session")
def series_small(net):
    """30 days of the synthetic series (SURVEY.md §8d generator, shorter horizon)."""
    from safe_marl_amd.series import make_synthetic_series
    return make_synthetic_series(net, n_days=30)


@pytest.fixture(scope="session")
def base_loads(net):
    buses = net["bus_numbers"]
    p = np.array([net["active_power_demand"][b] for b in buses])
    q = np.array([net["reactive_power_demand"][b] for b in buses])
    return p, q
