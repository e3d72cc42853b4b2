#!/usr/bin/env python3
"""tools/plan_probe.py -- time the FAST step for a list of launch plans (I,S,TILE) in ONE process, no torch.

    python3 tools/plan_probe.py [--bodies N] [--fp64] [--steps K] I,S,TILE [I,S,TILE ...]      (0,0,0 = automatic plan)
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bodies", type=int, default=262144)
    ap.add_argument("--fp64", action="store_true")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--mode", choices=["fast", "strict"], default="fast")
    ap.add_argument("--data", choices=["normal", "shell", "zeros"], default="normal", help="positions: standard normal / the reference's SHELL start-up / all bodies at the origin")
    ap.add_argument("--masses", choices=["equal", "varied", "species"], default="equal",
                    help="varied: every chunk takes the generic (mass-multiplying) loop; species: three contiguous blocks of equal masses (1, 0.3, 2.5)")
    ap.add_argument("plans", nargs="*", default=["0,0,0"])
    args = ap.parse_args()
    pkg = entry.load_package()
    lib = pkg.lib()
    pkg.check(lib.nb_set_device(0))
    dtype = np.float64 if args.fp64 else np.float32
    n = args.bodies
    rng = np.random.default_rng(1)
    pos = rng.standard_normal((n, 4)).astype(dtype)
    pos[:, 3] = 1
    vel = np.zeros((n, 4), dtype)
    if args.data == "shell":
        import importlib.util

        spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
        bench = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(bench)
        p0, v0 = bench.make_bodies(n, dtype)
        pos, vel = p0.reshape(n, 4), v0.reshape(n, 4)
    elif args.data == "zeros":
        pos[:, :3] = 0
    if args.masses == "varied":
        pos = pos.copy()
        pos[:, 3] = (0.5 + rng.random(n)).astype(dtype)
    if args.masses == "species":
        pos = pos.copy()
        pos[n // 3:, 3] = 0.3
        pos[2 * n // 3:, 3] = 2.5
    mode = pkg.NB_MODE_FAST if args.mode == "fast" else pkg.NB_MODE_STRICT
    system = pkg.BodySystemHIP(n, 256, pkg.NBodyParams(), dtype, pos.ravel(), vel.ravel(), mode=mode)
    flops = 30 if args.fp64 else 20
    peak = 78.6e12 if args.fp64 else 157.3e12
    for plan in args.plans:
        I, S, T = (int(x) for x in plan.split(","))
        pkg.set_plan_override(I, S, T)
        p = pkg.plan(n, n, dtype)
        for _ in range(2):
            system.update(dtype(0.016))
        e0, e1 = pkg.Event(), pkg.Event()
        system.synchronize()
        e0.record()
        for _ in range(args.steps):
            system.update(dtype(0.016))
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_ms(e1) / args.steps
        print(json.dumps({"data": args.data, "masses": args.masses, "plan": plan, "I": p.bodies_per_lane, "S": p.lanes_per_body, "tile": p.tile_bodies, "grid": p.grid_blocks, "lds": p.lds_bytes,
                          "ms": round(ms, 4), "ginter_per_s": round(n * n / ms * 1e-6, 1), "frac": round(flops * n * n / (ms * 1e-3) / peak, 4)}), flush=True)
    pkg.set_plan_override(0, 0, 0)


if __name__ == "__main__":
    main()
