#!/usr/bin/env python3
"""tools/pair_crossover.py -- where does the pairwise layout (nb_integrate_ws_*) start to win, and with which geometry?
For each body count: the one-sided FAST kernel (nb_integrate_*) against the pairwise layout under every plan
(R vectors per lane, S waves per workgroup, C workgroups per block); ms per step over a few hundred steps.

    python3 tools/pair_crossover.py [f32|f64] > gpurun_out/pair_crossover.txt
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

pkg = entry.load_package()
pkg.check(pkg.lib().nb_set_device(0))
dtype = np.float64 if (len(sys.argv) > 1 and sys.argv[1] == "f64") else np.float32
host = entry.load_oracle().Oracle()  # (start-up bodies only)


def time_system(n, pos0, vel0, workspace, steps):
    s = pkg.BodySystemHIP(n, 256, pkg.NBodyParams(), dtype, pos0, vel0, mode=pkg.NB_MODE_FAST, workspace=workspace)
    dt = dtype(np.float32(0.016))
    for _ in range(3):
        s.update(dt)
    s.synchronize()
    e0, e1 = pkg.Event(), pkg.Event()
    e0.record(None)
    for _ in range(steps):
        s.update(dt)
    e1.record(None)
    e1.synchronize()
    s.free()
    return e0.elapsed_ms(e1) / steps


sizes = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [4096, 8192, 12288, 16384, 24576, 32768, 49152, 65536, 98304, 131072]
for n in sizes:
    p32, v32 = host.startup_state(n, np.float32)
    pos0, vel0 = p32.astype(dtype), v32.astype(dtype)
    steps = max(20, min(400, int(2e10 / (float(n) * n))))
    one = time_system(n, pos0, vel0, False, steps)
    rows = []
    for R in (1, 2, 4):
        for S in (4, 8, 16):
            for C in (1, 2, 4, 8, 16, 32):
                pkg.set_pair_plan_override(R, S, C, 1)
                pl = pkg.pair_plan(n, dtype)
                units = (pl.blocks // 2 + 1) * pl.bodies_per_lane
                if units < C * S or pl.grid_blocks > 8192:  # a wave without a unit / far more workgroups than the chip can use
                    continue
                rows.append({"R": R, "S": S, "C": C, "ms": time_system(n, pos0, vel0, True, steps)})
    pkg.set_pair_plan_override(0, 0, 0, 0)
    auto = pkg.pair_plan(n, dtype)
    rows.sort(key=lambda r: r["ms"])
    print(json.dumps({"bodies": n, "dtype": np.dtype(dtype).name, "one_sided_ms": round(one, 5), "best": rows[:5], "worst": rows[-1],
                      "automatic_plan": {"applies": auto.applies, "R": auto.bodies_per_lane // (2 if dtype == np.float32 else 1), "S": auto.waves_per_block, "C": auto.splits}}), flush=True)
