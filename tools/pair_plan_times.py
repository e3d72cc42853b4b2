#!/usr/bin/env python3
"""tools/pair_plan_times.py -- ms per step of the pairwise layout at given sizes under given plans (R,S,C; 0,0,0 = automatic),
with the two kernels of the step timed separately (probe event), next to the one-sided kernel.

    python3 tools/pair_plan_times.py f32 65536,16384 0,0,0 4,16,2 4,8,4 [--reps 3]
"""
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
dtype = np.float64 if sys.argv[1] == "f64" else np.float32
sizes = [int(x) for x in sys.argv[2].split(",")]
plans = [tuple(int(v) for v in a.split(",")) for a in sys.argv[3:] if a.count(",") == 2]
reps = int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else 3
host = entry.load_oracle().Oracle()  # (start-up bodies only)


def time_system(n, pos0, vel0, workspace, steps):
    s = pkg.BodySystemHIP(n, 256, pkg.NBodyParams(), dtype, pos0, vel0, mode=pkg.NB_MODE_FAST, workspace=workspace)
    dt = dtype(np.float32(0.016))
    for _ in range(3):
        s.update(dt)
    s.synchronize()
    best = 1e30
    for _ in range(reps):
        e0, e1 = pkg.Event(), pkg.Event()
        e0.record(None)
        for _ in range(steps):
            s.update(dt)
        e1.record(None)
        e1.synchronize()
        best = min(best, e0.elapsed_ms(e1) / steps)
    split = None
    if workspace and s._workspace is not None:
        a, b, c = pkg.Event(), pkg.Event(), pkg.Event()
        pkg.check(lib.nb_set_pair_probe_event(b.h))
        f = g = 0.0
        for _ in range(20):
            a.record(None)
            s.update(dt)
            c.record(None)
            c.synchronize()
            f += a.elapsed_ms(b)
            g += b.elapsed_ms(c)
        pkg.check(lib.nb_set_pair_probe_event(None))
        split = (round(f / 20 * 1e3, 1), round(g / 20 * 1e3, 1))
    s.free()
    return best, split


for n in sizes:
    p32, v32 = host.startup_state(n, np.float32)
    pos0, vel0 = p32.astype(dtype), v32.astype(dtype)
    steps = max(10, min(400, int(4e10 / (float(n) * n))))
    one, _ = time_system(n, pos0, vel0, False, steps)
    peak = (78.6e12 / 30) if dtype == np.float64 else (157.3e12 / 20)
    row = {"bodies": n, "one_sided_us": round(one * 1e3, 1), "one_sided_frac": round(float(n) * n / (one * 1e-3) / peak, 4), "plans": []}
    for plan in plans:
        pkg.set_pair_plan_override(*plan, 1 if any(plan) else 0)
        pl = pkg.pair_plan(n, dtype)
        if not pl.applies:
            continue
        ms, split = time_system(n, pos0, vel0, True, steps)
        row["plans"].append({"plan": plan, "used": [pl.bodies_per_lane, pl.waves_per_block, pl.splits, pl.blocks], "us": round(ms * 1e3, 1), "frac": round(float(n) * n / (ms * 1e-3) / peak, 4),
                             "forces_finish_us": split})
    pkg.set_pair_plan_override(0, 0, 0, 0)
    print(json.dumps(row), flush=True)
