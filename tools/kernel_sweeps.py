#!/usr/bin/env python3
"""tools/kernel_sweeps.py -- tuning sweeps and one-rank projections on ONE GPU (moved out of bench.py in round 4: none of this is
the benchmark).  Everything goes through the C-ABI; the process-global plan overrides come from include/nbody_hip_tuning.h.

    python tools/kernel_sweeps.py --sweep [--bodies N] [--fp64]                   every one-sided FAST geometry at N bodies
    python tools/kernel_sweeps.py --emulate-gpus G [--layout pairwise|one-sided]  the kernels ONE rank of a G-rank step launches
    python tools/kernel_sweeps.py --emulate-gpus G --sweep                        ... for every one-sided geometry (tile schedule)

Each mode prints JSON lines of its own; none of them is the headline metric.  The emulations run a rank's kernel schedule alone
on the chip with no exchange: a projection of the compute side of a multi-GPU step, nothing more.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402
import bench  # noqa: E402  (make_bodies, the peaks)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bodies", type=int, default=262144)
    ap.add_argument("--fp64", action="store_true")
    ap.add_argument("--mode", choices=["fast", "strict"], default="fast")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--sweep", action="store_true")
    ap.add_argument("--emulate-gpus", type=int, default=0)
    ap.add_argument("--layout", choices=["pairwise", "one-sided"], default="pairwise")
    ap.add_argument("--schedule", choices=["tiles", "chunks"], default="tiles", help="one-sided emulation: one launch per position tile, or own/below/above")
    args = ap.parse_args()

    dtype = np.float64 if args.fp64 else np.float32
    n, G = args.bodies, args.emulate_gpus
    pos0, vel0 = bench.make_bodies(n, dtype)
    pkg = entry.load_package()
    lib = pkg.lib()
    pkg.check(lib.nb_set_device(0), "nb_set_device")
    mode = pkg.NB_MODE_FAST if args.mode == "fast" else pkg.NB_MODE_STRICT
    params = pkg.NBodyParams()
    dt, damping = dtype(np.float32(params.time_step)), dtype(np.float32(params.damping))
    soft = dtype(np.float32(params.softening))
    pkg.set_softening_squared(soft * soft if args.fp64 else np.float32(soft * soft))
    shard_fn = lib.nb_integrate_shard_f64 if args.fp64 else lib.nb_integrate_shard_f32
    flops, peak = (30, bench.FP64_VECTOR_PEAK_TFLOPS) if args.fp64 else (20, bench.FP32_VECTOR_PEAK_TFLOPS)
    old, new, vel, acc = (pkg.DeviceBuffer(pos0.nbytes) for _ in range(4))
    old.upload(pos0), vel.upload(vel0)

    def launch(i0, ni, j0, nj, flags):
        pkg.check(shard_fn(new.ptr, old.ptr, vel.ptr, acc.ptr, i0, ni, j0, nj, flags, dt, damping, 256, mode, None), "nb_integrate_shard")

    def timed(one_step, steps=args.steps, warmup=args.warmup):
        for _ in range(warmup):
            one_step()
        pkg.check(lib.nb_device_synchronize())
        e0, e1 = pkg.Event(), pkg.Event()
        e0.record(None)
        for _ in range(steps):
            one_step()
        e1.record(None)
        e1.synchronize()
        return e0.elapsed_ms(e1) / steps

    def geometries():
        for I in ((1, 2, 4) if args.fp64 else (2, 4)):
            for S in (4, 8, 16, 64):
                for tile in (256, 512, 1024, 2048):
                    blk = 256 if S == 64 else 64 * S
                    if tile < blk or tile // blk not in (1, 2, 4):
                        continue
                    if S == 64 and (tile not in (512, 1024) or I > (2 if args.fp64 else 4)):
                        continue
                    if lib.nb_set_plan_override(I, S, tile) == 0:
                        yield I, S, tile

    sharded = entry.load_package_module("sharded")

    def rank_schedule(r):
        i0, ni = sharded.slice_of(r, G, n)
        sched = sharded.tile_schedule(r, G, n, mode == pkg.NB_MODE_STRICT) if args.schedule == "tiles" else sharded.chunk_schedule(i0, ni, n, mode == pkg.NB_MODE_STRICT)
        return i0, ni, sched

    def one_sided_step(i0, ni, sched):
        for k, (j0, nj, _) in enumerate(sched):
            launch(i0, ni, j0, nj, (pkg.NB_SHARD_ACC_IN if k else 0) | (pkg.NB_SHARD_FINALIZE if k == len(sched) - 1 else 0))

    if G > 1 and args.sweep:  # every geometry for one rank's tile schedule
        i0, ni, sched = rank_schedule(G // 2)
        rows = []
        try:
            for I, S, tile in geometries():
                ms = timed(lambda: one_sided_step(i0, ni, sched))
                rows.append(dict(I=I, S=S, tile=tile, ms=round(ms, 4), speedup_vs_ideal=round((float(n) * n / G) / (ms * 1e-3) * 1e-12, 3)))
                print(json.dumps(rows[-1]), flush=True)
        finally:
            pkg.set_plan_override(0, 0, 0)
        print("best:", json.dumps(min(rows, key=lambda r: r["ms"])))
    elif G > 1 and args.layout == "pairwise" and mode == pkg.NB_MODE_FAST:
        # one rank's kernels of the pairwise step across G ranks (nb_emulate_pair_rank_*: diagonal, G/2 rectangles, folds, finish)
        emulate = lib.nb_emulate_pair_rank_f64 if args.fp64 else lib.nb_emulate_pair_rank_f32
        need = ctypes.c_size_t(0)
        pkg.check(emulate(None, None, None, None, ctypes.byref(need), n, G, 0, dt, damping, None), "nb_emulate_pair_rank (size)")
        work = pkg.DeviceBuffer(need.value)
        out = []
        for r in sorted({0, G // 2, G - 1}):
            ms = timed(lambda: pkg.check(emulate(new.ptr, old.ptr, vel.ptr, work.ptr, ctypes.byref(need), n, G, r, dt, damping, None), "nb_emulate_pair_rank"))
            out.append({"rank": r, "ms_per_step_kernels_only": ms, "launches_per_step": 2 * (G // 2) + 2})
        worst = max(o["ms_per_step_kernels_only"] for o in out)
        print(json.dumps({"emulated_gpus": G, "bodies": n, "schedule": "pairwise across ranks: diagonal + G/2 rectangles, reaction sums to their owners",
                          "workspace_bytes_per_rank": need.value, "ranks": out, "projected_interactions_per_s_excluding_exchange": float(n) * n / (worst * 1e-3)}), flush=True)
    elif G > 1:
        out = []
        for r in sorted({0, G // 2, G - 1}):
            i0, ni, sched = rank_schedule(r)
            ms = timed(lambda: one_sided_step(i0, ni, sched))
            pl = pkg.plan(ni, n, dtype)
            out.append({"rank": r, "ms_per_step_kernels_only": ms, "launches_per_step": len(sched), "plan": [pl.bodies_per_lane, pl.lanes_per_body, pl.tile_bodies, pl.grid_blocks]})
        worst = max(o["ms_per_step_kernels_only"] for o in out)
        print(json.dumps({"emulated_gpus": G, "bodies": n, "schedule": "tiles" if args.schedule == "tiles" else "own/below/above", "ranks": out,
                          "projected_interactions_per_s_excluding_exchange": float(n) * n / (worst * 1e-3)}), flush=True)
    elif args.sweep:
        rows = []
        try:
            for I, S, tile in geometries():
                ms = timed(lambda: launch(0, n, 0, n, pkg.NB_SHARD_FINALIZE), steps=5)
                rows.append(dict(I=I, S=S, tile=tile, ms=round(ms, 4), ginter=round(n * n / ms * 1e-6, 1), frac=round(flops * n * n / (ms * 1e-3) / (peak * 1e12), 4)))
                print(json.dumps(rows[-1]), flush=True)
        finally:
            pkg.set_plan_override(0, 0, 0)
        print("best:", json.dumps(max(rows, key=lambda r: r["ginter"])))
    else:
        raise SystemExit("nothing to do: --sweep and/or --emulate-gpus G")
    for b in (old, new, vel, acc):
        b.free()


if __name__ == "__main__":
    main()
