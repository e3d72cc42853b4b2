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


def load_golden_compact(n, tag, oracle):
    """shell_n{n}_{tag}_compact.npz (tests/golden/make_golden.py): the initial state is drawn by the oracle's randomise_bodies
    and checked against the fixture's SHA-256; the later states hold x, y, z only (mass 1, velocity .w 0 throughout).
    Returns a dict with full pos_0 / vel_0 / pos_k [/ vel_k] arrays of 4N values."""
    import hashlib

    raw = np.load(os.path.join(GOLDEN_DIR, f"shell_n{n}_{tag}_compact.npz"), allow_pickle=False)
    dtype = np.float32 if tag == "f32" else np.float64
    pos0, vel0 = oracle.startup_state(n, dtype)
    assert hashlib.sha256(pos0.tobytes()).digest() == raw["sha256_pos_0"].tobytes(), "start-up positions differ from the fixture's"
    assert hashlib.sha256(vel0.tobytes()).digest() == raw["sha256_vel_0"].tobytes(), "start-up velocities differ from the fixture's"
    out = {"pos_0": pos0, "vel_0": vel0}
    for key in raw.files:
        if not key.endswith("_xyz"):
            continue
        full = np.zeros((n, 4), dtype)
        full[:, :3] = raw[key]
        full[:, 3] = 1 if key.startswith("pos_") else 0
        out[key[:-4]] = full.reshape(-1)
    return out
