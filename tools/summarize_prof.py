#!/usr/bin/env python3
"""Condense a tools/profile.sh run (gpurun_out/prof) into the tracked files under profiles/.

  python tools/summarize_prof.py gpurun_out/prof profiles/round1

writes <prefix>_kernel_stats.csv (rocprofv3 --kernel-trace --stats, verbatim),
       <prefix>_pmc_summary.json (per-launch means of every counter for the dominant kernel + derived figures).
HBM traffic follows MI355X_MICROARCH.md "HBM": FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports
exactly half the bytes of a wide (16 B/lane) coalesced read stream, so reads = 2 * FETCH_SIZE * 1024;
WRITE_SIZE is exact for 16 B/lane stores.
"""
import collections
import csv
import glob
import json
import shutil
import sys

src, prefix = sys.argv[1], sys.argv[2]
KERNEL = sys.argv[3] if len(sys.argv) > 3 else "integrate_bodies_fast"

import os

def newest(pattern):
    """gpurun merges successive runs into the same local directory: take the most recent file"""
    return max(glob.glob(pattern), key=os.path.getmtime)

stats = newest(f"{src}/trace/*/*_kernel_stats.csv")
shutil.copy(stats, f"{prefix}_kernel_stats.csv")
def is_kernel(name):
    """the kernel asked for -- not its clock-reading twin (pair_forces_clocked: ten launches after bench.py's timed region)"""
    return KERNEL in name and (("_clocked" in KERNEL) or "_clocked" not in name)

kernel_row = next(r for r in csv.DictReader(open(stats)) if is_kernel(r["Name"]))

counters = {}
meta = {}
for f in [newest(os.path.join(d, "*", "*_counter_collection.csv")) for d in sorted(glob.glob(f"{src}/pmc_*")) if os.path.isdir(d)]:
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if is_kernel(r["Kernel_Name"]):
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta = {k: r[k] for k in ("Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Scratch_Size")}
    for k, v in agg.items():
        counters[k] = sum(v) / len(v)

avg_ns = float(kernel_row["AverageNs"])
out = {"kernel": meta, "launches_profiled": int(kernel_row["Calls"]), "kernel_trace_avg_ms": avg_ns * 1e-6,
       "kernel_trace_min_ms": float(kernel_row["MinNs"]) * 1e-6, "counters_per_launch_mean": counters, "derived": {}}
d = out["derived"]
if "FETCH_SIZE" in counters and "WRITE_SIZE" in counters:
    d["hbm_read_bytes_per_launch"] = 2 * counters["FETCH_SIZE"] * 1024
    d["hbm_write_bytes_per_launch"] = counters["WRITE_SIZE"] * 1024
    d["hbm_bytes_per_launch"] = d["hbm_read_bytes_per_launch"] + d["hbm_write_bytes_per_launch"]
    d["hbm_GBps"] = d["hbm_bytes_per_launch"] / (avg_ns * 1e-9) / 1e9
if "GRBM_GUI_ACTIVE" in counters:
    d["effective_clock_GHz"] = counters["GRBM_GUI_ACTIVE"] / 8 / avg_ns  # summed over the 8 XCDs
    cycles = counters["GRBM_GUI_ACTIVE"] / 8
    if "SQ_ACTIVE_INST_VALU" in counters:
        # SQ_ACTIVE_INST_* count quad-cycles summed over all SIMDs (256 CUs x 4); GRBM_GUI_ACTIVE comes from ANOTHER pass (another launch, another
        # clock): kept for the record only
        d["valu_busy_vs_grbm_of_another_pass"] = counters["SQ_ACTIVE_INST_VALU"] * 4 / (cycles * 1024)
if "SQ_ACTIVE_INST_VALU" in counters and "SQ_BUSY_CYCLES" in counters:
    # Numerator and denominator of ONE pass (round-5 review: the figure above divided by a GRBM_GUI_ACTIVE of another launch and read 1.011):
    # SQ_BUSY_CYCLES sums the busy cycles of the 32 shader engines (4 per XCD), SQ_ACTIVE_INST_VALU quad-cycles over the 1 024 SIMDs.  With two
    # waves per SIMD the windows in which each has a vector instruction active can overlap, so the raw quotient can pass 1: it is kept as
    # `valu_busy_raw`, and `valu_busy_fraction` says min(raw, 1) -- read a capped value as ">= 0.99".
    raw = counters["SQ_ACTIVE_INST_VALU"] * 4 / (counters["SQ_BUSY_CYCLES"] / 32 * 1024)
    d["valu_busy_raw"] = raw
    d["valu_busy_fraction"] = min(raw, 1.0)
    d["valu_busy_counts"] = "SQ_ACTIVE_INST_VALU x 4 / (SQ_BUSY_CYCLES / 32 x 1024 SIMDs), same --pmc pass; capped at 1 (the active windows of a SIMD's two waves overlap)"
if "SQ_INSTS_VALU" in counters:
    d["valu_wave_instructions_per_launch"] = counters["SQ_INSTS_VALU"]
# the launch plan and commit the counters belong to (bench.py only quotes `traffic` from a summary whose plan matches its own)
out["mode"] = "strict" if "strict" in KERNEL else "fast"
try:
    with open(os.path.join(src, "trace.log")) as fh:
        line = next(l for l in reversed(fh.read().splitlines()) if l.startswith("{") and "kernel_plan" in l)
    bench_line = json.loads(line)
    out["workload"] = bench_line["config"]["workload"]
    # nb_plan_* describes the FAST geometry only: it says nothing about a STRICT launch
    out["kernel_plan"] = bench_line["config"]["kernel_plan"] if out["mode"] == "fast" else None
except (OSError, StopIteration, KeyError, ValueError):
    out["kernel_plan"] = None
out["notes"] = ("kernel.VGPR_Count / LDS_Block_Size are rocprofv3's dispatch-packet fields as it prints them (VGPR_Count in allocation granules as "
                "reported by the tool, LDS_Block_Size without the dynamic part requested at launch); the register and LDS figures quoted in "
                "DESIGN.md come from the compiled ISA (make -C cuda-nbody_amd/csrc asm) and from nb_plan_*")
json.dump(out, open(f"{prefix}_pmc_summary.json", "w"), indent=1)
print(json.dumps(out["derived"], indent=1))
print("kernel avg ms", out["kernel_trace_avg_ms"])
