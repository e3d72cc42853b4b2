#!/usr/bin/env python3
"""tools/stamp_probe.py -- with a diagnostic build of the library (make -C cuda-nbody_amd/csrc EXP=-DNB_STAMPS OUT=...),
record when each wave of each workgroup of the FAST kernel starts and stops streaming (s_memrealtime, 10 ns ticks), and print how the
waves of a workgroup spread out: the under-filled tail of a workgroup is (last finish - mean finish).

    NBODY_HIP_LIB=/path/libnbody_hip_stamps.so python3 tools/stamp_probe.py [--bodies N] [--plan I,S,TILE]
"""
import argparse
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--bodies", type=int, default=262144)
ap.add_argument("--plan", default="4,16,2048")
args = ap.parse_args()
pkg = entry.load_package()
lib = pkg.lib()
pkg.check(lib.nb_set_device(0))
n = args.bodies
I, S, T = (int(x) for x in args.plan.split(","))
pkg.set_plan_override(I, S, T)
plan = pkg.plan(n, n, np.float32)
rng = np.random.default_rng(1)
pos = rng.standard_normal((n, 4)).astype(np.float32)
pos[:, 3] = 1
bufs = [pkg.DeviceBuffer(pos.nbytes) for _ in range(4)]  # old, new, vel, acc(stamps)
bufs[0].upload(pos.ravel())
pkg.check(lib.nb_set_softening_sq_f32(np.float32(0.01)))
for _ in range(3):
    pkg.check(lib.nb_integrate_shard_f32(bufs[1].ptr, bufs[0].ptr, bufs[2].ptr, bufs[3].ptr, 0, n, 0, n, pkg.NB_SHARD_FINALIZE, 0.016, 1.0, 256, pkg.NB_MODE_FAST, None))
pkg.check(lib.nb_device_synchronize())
raw = bufs[3].download(np.zeros(4 * n, np.float32)).view(np.uint64)
waves = plan.block_threads // 64
st = raw[:plan.grid_blocks * waves * 2].reshape(plan.grid_blocks, waves, 2).astype(np.float64)
t0 = st[:, :, 0].min(axis=1, keepdims=True)
dur = st[:, :, 1] - t0                       # per wave: finish time since the workgroup's first start
span = dur.max(axis=1)                       # workgroup busy span
print(f"plan {args.plan}: {plan.grid_blocks} workgroups x {waves} waves; median workgroup span {np.median(span) * 0.01:.1f} us")
rel = np.sort(dur / span[:, None], axis=1)   # sorted finish times relative to the span
print("finish time of the k-th wave / workgroup span, median over workgroups:")
print("  " + " ".join(f"{x:.3f}" for x in np.median(rel, axis=0)))
print(f"mean finish / span = {np.median(rel.mean(axis=1)):.4f}  -> under-filled tail ~ {100 * (1 - np.median(rel.mean(axis=1))):.1f} % of the workgroup's time (if the SIMD has nothing else to run)")
# per SIMD mates (waves w, w+4, ...): spread inside a SIMD
