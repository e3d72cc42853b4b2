#!/usr/bin/env python3
"""tools/pair_stamps.py N [R,S,C] -- when do the waves of pair_forces start and finish?  Needs the diagnostic build
(tools/build_pair_variant.sh stamps "-DNB_PAIR_STAMPS"; run with NBODY_HIP_LIB=expv/libnbody_hip_stamps.so).  Prints, in microseconds
relative to the first wave's start (s_memtime ticks at 100 MHz): percentiles of start, loop entry, loop exit; per-SIMD spread of the
exit times; how many waves per (CU, SIMD)."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

pkg = entry.load_package()
lib = pkg.lib()
pkg.check(lib.nb_set_device(0))
n = int(sys.argv[1])
plan = tuple(int(x) for x in sys.argv[2].split(",")) if len(sys.argv) > 2 else (0, 0, 0)
host = entry.load_oracle().Oracle()
pos0, vel0 = host.startup_state(n, np.float32)
pkg.set_pair_plan_override(*plan, 1 if any(plan) else 0)
pl = pkg.pair_plan(n, np.float32)
s = pkg.BodySystemHIP(n, 256, pkg.NBodyParams(), np.float32, pos0, vel0, mode=pkg.NB_MODE_FAST, workspace=True)
for _ in range(20):
    s.update(np.float32(0.016))
s.synchronize()
waves = pl.grid_blocks * pl.waves_per_block
raw = np.zeros(8192 * 8, np.uint64)
fn = ctypes.CDLL(pkg.LIB_PATH).nb_debug_read_pair_stamps
fn.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
assert fn(raw.ctypes.data_as(ctypes.c_void_p), raw.nbytes) == 0
st = raw.reshape(8192, 8)[:min(waves, 8192)]
hw = (st[:, 3] & 0xffff).astype(int)
# s_memtime counters have different origins in different parts of the chip: take times relative to the first wave of the same CU
xcc_reg = (st[:, 3] >> 32).astype(int)
cu_key = xcc_reg * 1000 + ((hw >> 13) & 7) * 100 + ((hw >> 8) & 15)
print("CUs seen:", len(np.unique(cu_key)), " XCC_ID values:", sorted(set(xcc_reg.tolist()))[:10])
xcc = cu_key
t0 = np.zeros(len(st), np.uint64)
for x in np.unique(xcc):
    t0[xcc == x] = st[xcc == x, 0].min()
a, b, c = pkg.Event(), pkg.Event(), pkg.Event()
pkg.check(lib.nb_set_pair_probe_event(b.h))
a.record(None)
s.update(np.float32(0.016))
c.record(None)
c.synchronize()
pkg.check(lib.nb_set_pair_probe_event(None))
kernel_us = a.elapsed_ms(b) * 1e3
span = max(float((st[xcc == x, 2] - t0[xcc == x]).max()) for x in np.unique(xcc))
tick_us = kernel_us / span
spans = np.array([int((st[xcc == x, 2] - t0[xcc == x]).max()) for x in np.unique(xcc)])
print(f"pair_forces {kernel_us:.1f} us by HIP events; CU spans in ticks min/median/max {spans.min()} {int(np.median(spans))} {spans.max()} -> {1 / tick_us:.0f} ticks per us")
start, loop, done = ((st[:, k] - t0).astype(np.float64) * tick_us for k in range(3))
units = (st[:, 4] & 0xffffffff).astype(int)
simd, cu, se = (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 13) & 7
where = xcc * 100000 + se * 1000 + cu * 10 + simd
pct = lambda a: [round(float(np.percentile(a, p)), 1) for p in (0, 10, 50, 90, 100)]  # noqa: E731
print(f"{n} bodies plan {plan} -> I={pl.bodies_per_lane} S={pl.waves_per_block} C={pl.splits} blocks={pl.blocks}: {waves} waves")
print("start      us (min/10/50/90/max):", pct(start))
print("loop entry us:", pct(loop), " prologue us:", pct(loop - start))
print("loop exit  us:", pct(done), " units per wave:", sorted(set(units.tolist())))
ids, counts = np.unique(where, return_counts=True)
print("SIMDs in use:", len(ids), " waves per SIMD (min/median/max):", counts.min(), int(np.median(counts)), counts.max())
per_simd_units = np.array([units[where == i].sum() for i in ids])
per_simd_last = np.array([done[where == i].max() for i in ids])
per_simd_first = np.array([done[where == i].min() for i in ids])
print("units per SIMD (min/median/max):", per_simd_units.min(), int(np.median(per_simd_units)), per_simd_units.max())
print("last exit per SIMD us:", pct(per_simd_last), " first exit per SIMD us:", pct(per_simd_first))
print("spread of exits inside a SIMD us:", pct(per_simd_last - per_simd_first))
busy = per_simd_last - np.array([start[where == i].min() for i in ids])
print("us per unit and SIMD (last exit - first start) / units:", pct(busy / per_simd_units))
# which SIMD does wave w of a workgroup land on?  (the quarters of pair_forces give quarter p to a wave with w % 4 == p: one per SIMD if w -> w % 4)
wave_in_wg = (st[:, 4] >> 32).astype(int)
table = {}
for w, sd in zip(wave_in_wg.tolist(), simd.tolist()):
    table.setdefault(w, {}).setdefault(sd, 0)
    table[w][sd] += 1
print("SIMD by wave of the workgroup (wave: {simd: count}):", {w: table[w] for w in sorted(table)})
distinct = [len({int(simd[i]) for i in range(b, b + 4)}) for b in range(0, len(simd) - 3, pl.waves_per_block) if wave_in_wg[b] == 0]
print("workgroups whose waves 0-3 sit on four different SIMDs:", sum(1 for d in distinct if d == 4), "of", len(distinct))
