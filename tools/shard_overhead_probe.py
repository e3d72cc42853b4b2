#!/usr/bin/env python3
"""tools/shard_overhead_probe.py -- where does the time of one shard-sized FAST launch go?  With the stamps build
(tools/build_exp.sh stamps "-DNB_STAMPS"): per-wave streaming start/finish (s_memtime, one clock for the whole chip) against the
launch's duration from HIP events: time before the first wave streams, streaming span, time after the last wave finished.

    NBODY_HIP_LIB=exp/libnbody_hip_stamps.so python3 tools/shard_overhead_probe.py [--bodies 32768] [--plan 2,16,2048]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--bodies", type=int, default=32768)
ap.add_argument("--plan", default="2,16,2048")
args = ap.parse_args()
pkg = entry.load_package()
lib = pkg.lib()
pkg.check(lib.nb_set_device(0))
n = args.bodies
pkg.set_plan_override(*(int(x) for x in args.plan.split(",")))
plan = pkg.plan(n, n, np.float32)
rng = np.random.default_rng(1)
pos = rng.standard_normal((n, 4)).astype(np.float32)
pos[:, 3] = 1
bufs = [pkg.DeviceBuffer(max(pos.nbytes, plan.grid_blocks * (plan.block_threads // 64) * 16)) for _ in range(4)]  # old, new, vel, acc(stamps)
bufs[0].upload(pos.ravel())
pkg.check(lib.nb_set_softening_sq_f32(np.float32(0.01)))
e0, e1 = pkg.Event(), pkg.Event()
ms = []
for rep in range(5):  # back to back, so that the chip holds its clock: only the LAST launch of a burst is timed (and its stamps read)
    for k in range(40):
        if k == 39:
            e0.record()
        pkg.check(lib.nb_integrate_shard_f32(bufs[1].ptr, bufs[0].ptr, bufs[2].ptr, bufs[3].ptr, 0, n, 0, n, pkg.NB_SHARD_FINALIZE, 0.016, 1.0, 256, pkg.NB_MODE_FAST, None))
    e1.record()
    e1.synchronize()
    ms.append(e0.elapsed_ms(e1))
raw = bufs[3].download(np.zeros(bufs[3].nbytes // 4, np.float32)).view(np.uint64)
waves = plan.block_threads // 64
st = raw[:plan.grid_blocks * waves * 2].reshape(plan.grid_blocks, waves, 2).astype(np.float64)
first, last_start, first_end, last = st[:, :, 0].min(), st[:, :, 0].max(), st[:, :, 1].min(), st[:, :, 1].max()
tick_ns = 10.0  # s_memtime counts at 100 MHz on this chip
print(f"bodies {n} plan {args.plan}: {plan.grid_blocks} workgroups x {waves} waves; launch (events, last of a burst of 40, median of 5) {np.median(ms) * 1e3:.1f} us")
print(f"  streaming: first wave starts at 0, last wave starts at {(last_start - first) * tick_ns * 1e-3:.1f} us, first wave finishes at {(first_end - first) * tick_ns * 1e-3:.1f} us, last at {(last - first) * tick_ns * 1e-3:.1f} us")
per_wave = (st[:, :, 1] - st[:, :, 0]) * tick_ns * 1e-3
print(f"  per-wave streaming time: median {np.median(per_wave):.1f} us, min {per_wave.min():.1f}, max {per_wave.max():.1f}")
print(f"  => outside the streaming span (launch, prologue of the first / epilogue of the last workgroup): {np.median(ms) * 1e3 - (last - first) * tick_ns * 1e-3:.1f} us")
