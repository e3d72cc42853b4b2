#!/usr/bin/env python3
"""tools/fast_error_table.py -- what FAST really does over 1 / 10 / 100 steps: max, 99th percentile and median of the
per-body relative position error against the CPU path's trajectory (tests/golden, the oracle's output), N = 8 ... 4096,
fp32 and fp64.  Prints one JSON line per (N, precision); DESIGN.md section 5 quotes it.

    python3 tools/fast_error_table.py > gpurun_out/fast_vs_cpu_path_errors.jsonl
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as entry  # noqa: E402
from conftest import golden_steps, load_golden, xyz  # noqa: E402

pkg = entry.load_package()
pkg.check(pkg.lib().nb_set_device(0))
for n in (8, 256, 1024, 4096):
    for tag, dtype in (("f32", np.float32), ("f64", np.float64)):
        g = load_golden(n, tag)
        system = pkg.BodySystemHIP(n, 256, pkg.NBodyParams(), dtype, g["pos_0"], g["vel_0"], mode=pkg.NB_MODE_FAST)
        row, done = {"bodies": n, "precision": tag}, 0
        for s in golden_steps(g):
            for _ in range(s - done):
                system.update(dtype(np.float32(0.016)))
            done = s
            a, b = xyz(system.get_position()).astype(np.float64), xyz(g[f"pos_{s}"]).astype(np.float64)
            err = np.linalg.norm(a - b, axis=1) / np.linalg.norm(b, axis=1)
            row[f"steps_{s}"] = {"max": float(err.max()), "p99": float(np.percentile(err, 99)), "median": float(np.median(err))}
        system.free()
        print(json.dumps(row), flush=True)
