"""A refused allocation must not poison what follows it.

On ROCm hipGetLastError returns the LAST error of the thread and only then clears it; every launcher of the library ends with
`launch; return hipGetLastError();`.  So a hipMalloc that failed -- the event the out-of-memory fall-backs are built on (halve the
workspace and ask again, step without one, `any shard fails, all shards lend nothing`) -- used to be reported once more as the
status of the next, perfectly good, launch, and the fall-back threw instead of falling back.  Since round 5 nb_alloc clears the
thread's error after handing the status to its caller and every launcher discards a stale one before it launches.  The refusals
here are REAL ones: nb_set_alloc_limit / --alloc-limit-mib replace a request above the limit by one no device can serve.  That hook
belongs to the lab library (include/nbody_hip_lab.h, round 6): the in-process check runs in a child that loads libnbody_hip_lab.so,
the CLI runs preload it (LD_PRELOAD) -- the same object files as libnbody_hip.so plus the lab's."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "cuda-nbody_amd", "nbody")
LAB_LIB = os.path.join(ROOT, "cuda-nbody_amd", "libnbody_hip_lab.so")


def _with_lab(env=None):
    """the environment of a CLI run whose nb_* calls land in the lab library (it exports everything the product does + the hooks)"""
    env = dict(os.environ if env is None else env)
    env["LD_PRELOAD"] = LAB_LIB + ((":" + env["LD_PRELOAD"]) if env.get("LD_PRELOAD") else "")
    return env


LIMIT_HOOK_CHECK = """
import ctypes, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
import __graft_entry__ as entry
gpu = entry.load_package()
gpu.use_lab()
lib = gpu.lib()
gpu.check(lib.nb_set_device(0))
n = 16384
pos0, vel0 = entry.load_oracle().Oracle().startup_state(n, np.float32)
gpu.set_softening_squared(np.float32(0.1) * np.float32(0.1))
bufs = [gpu.DeviceBuffer(pos0.nbytes) for _ in range(3)]
bufs[0].upload(pos0), bufs[2].upload(vel0)
need = gpu.workspace_bytes(n, np.float32)
work = gpu.DeviceBuffer(need)
args = (np.float32(0.016), np.float32(1), n, 256, gpu.NB_MODE_FAST)
gpu.check(lib.nb_set_alloc_limit(1 << 20))
p = ctypes.c_void_p()
assert lib.nb_alloc(ctypes.byref(p), (1 << 20) + 1) == gpu.NB_ERR_OUT_OF_MEMORY and not p.value
assert lib.nb_integrate_ws_f32(bufs[1].ptr, bufs[0].ptr, bufs[2].ptr, *args, work.ptr, need, None) == 0
assert lib.nb_alloc(ctypes.byref(p), 1 << 20) == 0 and p.value
gpu.check(lib.nb_free(p))
gpu.check(lib.nb_set_alloc_limit(0))
gpu.check(lib.nb_device_synchronize())
assert np.isfinite(bufs[1].download(np.zeros_like(pos0))).all()
print("limit hook ok")
"""


@pytest.mark.gpu
def test_steps_after_a_refused_allocation_succeed(gpu, oracle):
    lib = gpu.lib()
    n = 16384
    pos0, vel0 = oracle.startup_state(n, np.float32)
    gpu.set_softening_squared(np.float32(0.1) * np.float32(0.1))
    bufs = [gpu.DeviceBuffer(pos0.nbytes) for _ in range(3)]
    bufs[0].upload(pos0), bufs[2].upload(vel0)
    need = gpu.workspace_bytes(n, np.float32)
    work = gpu.DeviceBuffer(need)
    args = (np.float32(0.016), np.float32(1), n, 256, gpu.NB_MODE_FAST)

    def refused():
        p = ctypes.c_void_p()
        rc = lib.nb_alloc(ctypes.byref(p), 1 << 60)
        assert rc == gpu.NB_ERR_OUT_OF_MEMORY and not p.value and b"emory" in lib.nb_error_string(rc)  # hipErrorOutOfMemory: the value the host turns into std::bad_alloc
        return rc

    refused()
    assert lib.nb_integrate_f32(bufs[1].ptr, bufs[0].ptr, bufs[2].ptr, *args, None) == 0
    refused()
    assert lib.nb_integrate_ws_f32(bufs[0].ptr, bufs[1].ptr, bufs[2].ptr, *args, work.ptr, need, None) == 0
    refused()
    assert lib.nb_integrate_f32(bufs[1].ptr, bufs[0].ptr, bufs[2].ptr, np.float32(0.016), np.float32(1), n, 256, gpu.NB_MODE_STRICT, None) == 0
    # the same through the limit hook (what the CLI tests below rely on): a request above it is refused by the runtime, one below
    # passes -- in a child process, on the lab library
    done = subprocess.run([sys.executable, "-c", LIMIT_HOOK_CHECK, ROOT], capture_output=True, text=True, timeout=300)
    assert done.returncode == 0 and "limit hook ok" in done.stdout, done.stderr[-3000:]
    gpu.check(lib.nb_device_synchronize())
    got = bufs[1].download(np.zeros_like(pos0))
    assert np.isfinite(got).all()
    for b in bufs + [work]:
        b.free()


def _dump(tmp_path, name, *flags, env=None):
    out = tmp_path / f"{name}.bin"
    r = subprocess.run([CLI, "--numbodies=262144", "--steps=2", f"--dump={out}", *flags], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    return np.fromfile(out, dtype=np.float32)


@pytest.mark.gpu
def test_cli_halves_the_workspace_when_the_device_refuses_it(tmp_path):
    """BodySystemHIPStored::ensure_workspace (host/bodysystemhip_storage.cpp): 262 144 bodies ask for 384 MiB of reaction slots.
    With allocations above 200 MiB refused, the body system must ask again for a form that fits in half (192 MiB: the tournament
    in slices, 105 MiB) and step with it -- the very bits of `--workspace-mib=192`; with everything above 60 MiB refused, nothing
    fits and the step is the one-sided kernel -- the very bits of `--no-workspace`.  Before round 5 both runs ended in
    "hipErrorOutOfMemory" thrown by the first launch AFTER the refused allocation."""
    capped = _dump(tmp_path, "capped", "--workspace-mib=192")
    one_sided = _dump(tmp_path, "one_sided", "--no-workspace")
    whole = _dump(tmp_path, "whole")
    assert capped.tobytes() != whole.tobytes() != one_sided.tobytes()
    assert _dump(tmp_path, "refused_once", "--alloc-limit-mib=200", env=_with_lab()).tobytes() == capped.tobytes()
    assert _dump(tmp_path, "refused_always", "--alloc-limit-mib=60", env=_with_lab()).tobytes() == one_sided.tobytes()
    # without the lab library in the process the CLI says what the flag needs instead of ignoring it
    r = subprocess.run([CLI, "--numbodies=1024", "--steps=1", "--alloc-limit-mib=60"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "libnbody_hip_lab.so" in (r.stdout + r.stderr)


@pytest.mark.gpu
def test_cli_shards_lend_nothing_when_one_allocation_is_refused(tmp_path):
    """BodySystemHIPSharded: when the workspace of any shard cannot be had, every shard lends nothing and the step is the one-sided
    tile schedule on all of them (the layout is the communicator's) -- the bits of `--no-workspace`, not an exception."""
    from test_comm_fake_rccl import _env

    env = _env()
    plain = _dump(tmp_path, "plain", "--devices=0,0,0,0", "--no-workspace", env=env)
    lent = _dump(tmp_path, "lent", "--devices=0,0,0,0", env=env)
    refused = _dump(tmp_path, "refused", "--devices=0,0,0,0", "--alloc-limit-mib=8", env=_with_lab(env))  # (the body arrays are 4 MiB each)
    assert lent.tobytes() != plain.tobytes() and refused.tobytes() == plain.tobytes()


@pytest.mark.gpu
def test_cli_leaves_cleanly_when_the_body_arrays_themselves_are_refused(tmp_path):
    """The body arrays are the caller's to allocate; when the device refuses THEM there is nothing to fall back to: std::bad_alloc,
    exit code 3 as in the reference's main (nbody.cpp:396-408) -- for the sharded system too, whose constructor has by then made its
    communicators and streams and must take them down again (BodySystemHIPSharded::allocate), not hang or abort in their destructors."""
    from test_comm_fake_rccl import _env

    for flags, env in ((("--numbodies=262144",), None), (("--numbodies=262144", "--devices=0,0,0,0"), _env())):
        r = subprocess.run([CLI, *flags, "--steps=1", "--alloc-limit-mib=2", f"--dump={tmp_path / 'never.bin'}"], capture_output=True, text=True, timeout=300, env=_with_lab(env))
        assert r.returncode == 3, (flags, r.returncode, r.stdout[-800:], r.stderr[-1500:])
        assert not (tmp_path / "never.bin").exists()
