#!/usr/bin/env python3
"""tools/pair_shard_projection.py -- what the pairwise step across ranks costs in KERNEL time, on one GPU.

G logical ranks share device 0 through the RCCL test double (tests/fake_rccl; ranks may share a device there), each with its own
stream, arrays and workspace, driven by nb_sharded_step_all_*.  All ranks' kernels run on the one GPU, so a step takes about the
SUM of the ranks' kernel times (+ the copies of the stand-in transport): step time / G is one rank's share -- a projection of
the compute side of a G-GPU step, not a measurement of one (no xGMI, no RCCL kernels).

    NBODY_RCCL_LIB=tests/fake_rccl/libfake_rccl.so python3 tools/pair_shard_projection.py [bodies] [f32|f64]
"""
import ctypes
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("NBODY_RCCL_LIB", os.path.join(ROOT, "tests", "fake_rccl", "libfake_rccl.so"))
import __graft_entry__ as entry  # noqa: E402

pkg = entry.load_package()
lib = pkg.lib()
pkg.check(lib.nb_set_device(0))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
dtype = np.float64 if (len(sys.argv) > 2 and sys.argv[2] == "f64") else np.float32
f32 = dtype == np.float32
host = entry.load_oracle().Oracle()  # (start-up bodies only)
p32, v32 = host.startup_state(n, np.float32)
pos0, vel0 = p32.astype(dtype), v32.astype(dtype)
soft = dtype(np.float32(0.1))
pkg.check(lib.nb_set_softening_sq_f32(np.float32(soft * soft)) if f32 else lib.nb_set_softening_sq_f64(float(soft * soft)))
step_all = lib.nb_sharded_step_all_f32 if f32 else lib.nb_sharded_step_all_f64
ws_bytes = lib.nb_comm_workspace_bytes_f32 if f32 else lib.nb_comm_workspace_bytes_f64
dt, one = dtype(np.float32(0.016)), dtype(1)


def run(G, workspace, steps):
    comms = (ctypes.c_void_p * G)()
    pkg.check(lib.nb_comm_init_all(comms, G, (ctypes.c_int * G)(*([0] * G))), "nb_comm_init_all")
    bufs, streams, works = [], [], []
    for k in range(G):
        b = [pkg.DeviceBuffer(pos0.nbytes) for _ in range(4)]
        b[0].upload(pos0), b[2].upload(vel0)
        bufs.append(b)
        s = ctypes.c_void_p()
        pkg.check(lib.nb_stream_create(ctypes.byref(s)))
        streams.append(s)
        if workspace:
            need = ctypes.c_size_t(0)
            pkg.check(ws_bytes(comms[k], n, pkg.NB_MODE_FAST, ctypes.byref(need)))
            if need.value:
                w = pkg.DeviceBuffer(need.value)
                works.append(w)
                pkg.check(lib.nb_comm_set_workspace(comms[k], w.ptr, need.value))
    arr = lambda xs: (ctypes.c_void_p * G)(*xs)  # noqa: E731
    read = 0

    def step():
        nonlocal read
        pkg.check(step_all(comms, G, arr([b[1 - read].ptr for b in bufs]), arr([b[read].ptr for b in bufs]), arr([b[2].ptr for b in bufs]), arr([b[3].ptr for b in bufs]),
                           n, dt, one, 256, pkg.NB_MODE_FAST, arr(streams)), "nb_sharded_step_all")
        read = 1 - read

    for _ in range(2):
        step()
    pkg.check(lib.nb_device_synchronize())
    e0, e1 = pkg.Event(), pkg.Event()
    e0.record(None)  # (the null stream orders against nothing here: time with the host clock around a device synchronize instead)
    import time
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    pkg.check(lib.nb_device_synchronize())
    ms = (time.perf_counter() - t0) / steps * 1e3
    for c in comms:
        pkg.check(lib.nb_comm_destroy(c))
    for b in bufs:
        for x in b:
            x.free()
    for w in works:
        w.free()
    return ms


steps = 10 if n <= 262144 else 3
single_pair = None
for G in (1, 2, 4, 8):
    one_sided = run(G, False, steps)
    pairwise = run(G, True, steps)
    if G == 1:
        single_pair = pairwise
    print(json.dumps({"bodies": n, "dtype": np.dtype(dtype).name, "ranks_on_one_gpu": G, "one_sided_step_ms": round(one_sided, 3), "pairwise_step_ms": round(pairwise, 3),
                      "per_rank_share_ms": {"one_sided": round(one_sided / G, 3), "pairwise": round(pairwise / G, 3)},
                      "projected_speedup_vs_pairwise_single_gpu": {"one_sided": round(single_pair / (one_sided / G), 2), "pairwise": round(single_pair / (pairwise / G), 2)}}), flush=True)
