import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as entry  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def O():
    """The oracle module (test infrastructure)."""
    mod = entry.load_oracle()
    mod.build()
    return mod


@pytest.fixture(scope="session")
def oracle(O):
    return O.Oracle()


@pytest.fixture(scope="session")
def pkg():
    return entry.load_package()


@pytest.fixture(scope="session")
def gpu(pkg):
    """Loads libnbody_hip.so and selects device 0; fails (does not skip) when the extension is missing."""
    pkg.lib()
    assert pkg.device_count() >= 1, "no HIP device visible"
    pkg.check(pkg.lib().nb_set_device(0), "nb_set_device")
    return pkg


GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def load_golden(n, tag):
    return np.load(os.path.join(GOLDEN_DIR, f"shell_n{n}_{tag}.npz"), allow_pickle=False)


def golden_steps(g):
    """the step counts a fixture holds besides 0 (N = 4096 stops at 10 to stay small)"""
    return sorted(int(k[4:]) for k in g.files if k.startswith("pos_") and k != "pos_0")


def xyz(a):
    return a.reshape(-1, 4)[:, :3]
