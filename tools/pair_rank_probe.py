#!/usr/bin/env python3
"""tools/pair_rank_probe.py -- kernel time of ONE rank of a G-rank step, on one GPU, no exchange: the one-sided tile schedule
(G launches of nb_integrate_shard_*, what round 2 ran) against the pairwise schedule across ranks (nb_emulate_pair_rank_*:
diagonal + G/2 rectangles + their folds + finish).  Projection of the compute side of strong scaling, not a measurement of it.

    python3 tools/pair_rank_probe.py [bodies] [f32|f64] [R S C]      (R S C: override the tiles' geometry, 0 = automatic)
"""
import ctypes
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
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
shard = lib.nb_integrate_shard_f32 if f32 else lib.nb_integrate_shard_f64
emulate = lib.nb_emulate_pair_rank_f32 if f32 else lib.nb_emulate_pair_rank_f64
dt, one = dtype(np.float32(0.016)), dtype(1)
if len(sys.argv) > 5:
    pkg.set_pair_plan_override(0, 0, 0, 0)
bufs = [pkg.DeviceBuffer(pos0.nbytes) for _ in range(4)]
bufs[0].upload(pos0), bufs[2].upload(vel0)
reps = 20 if n <= 262144 else 4


def timed(fn):
    for _ in range(2):
        fn()
    pkg.check(lib.nb_device_synchronize())
    e0, e1 = pkg.Event(), pkg.Event()
    e0.record(None)
    for _ in range(reps):
        fn()
    e1.record(None)
    e1.synchronize()
    return e0.elapsed_ms(e1) / reps


single = pkg.BodySystemHIP(n, 256, pkg.NBodyParams(), dtype, pos0, vel0, mode=pkg.NB_MODE_FAST, workspace=True)
single_ms = timed(lambda: single.update(dt))
single.free()
print(json.dumps({"bodies": n, "dtype": np.dtype(dtype).name, "single_gpu_pairwise_ms": round(single_ms, 4)}), flush=True)
if len(sys.argv) > 5:
    pkg.set_pair_plan_override(int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), 0)
    print("tile geometry override R S C =", sys.argv[3:6], flush=True)
for G in (2, 4, 8):
    ni = n // G
    need = ctypes.c_size_t(0)
    pkg.check(emulate(None, None, None, None, ctypes.byref(need), n, G, 0, dt, one, None), "size query")
    work = pkg.DeviceBuffer(need.value)
    rows = []
    for r in sorted({0, G // 2, G - 1}):
        def one_sided():
            for t in range(G):
                peer = (r + t) % G
                flags = (pkg.NB_SHARD_ACC_IN if t else 0) | (pkg.NB_SHARD_FINALIZE if t == G - 1 else 0)
                pkg.check(shard(bufs[1].ptr, bufs[0].ptr, bufs[2].ptr, bufs[3].ptr, r * ni, ni, peer * ni, ni, flags, dt, one, 256, pkg.NB_MODE_FAST, None), "nb_integrate_shard")

        def pairwise():
            pkg.check(emulate(bufs[1].ptr, bufs[0].ptr, bufs[2].ptr, work.ptr, ctypes.byref(need), n, G, r, dt, one, None), "nb_emulate_pair_rank")

        rows.append({"rank": r, "one_sided_ms": round(timed(one_sided), 4), "pairwise_ms": round(timed(pairwise), 4)})
    work.free()
    worst_one, worst_pair = max(x["one_sided_ms"] for x in rows), max(x["pairwise_ms"] for x in rows)
    print(json.dumps({"ranks": G, "bodies_per_rank": ni, "workspace_MiB_per_rank": round(need.value / 2**20, 1), "per_rank": rows,
                      "projected_speedup_vs_single_gpu_pairwise_excluding_exchange": {"one_sided": round(single_ms / worst_one, 2), "pairwise": round(single_ms / worst_pair, 2)}}), flush=True)
