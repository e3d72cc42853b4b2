#!/usr/bin/env python3
"""Condense a tools/ab_bench.sh file: per library and round ms per step, the two kernels of the pairwise step, the clock the chip delivered
(nb_clock_probe_launch) and hwmon's figure, socket power, interactions per joule; then the medians per library and B / A ratios."""
import json
import statistics
import sys

rows = [json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")]
by = {}
print(f"{'lib':34s} {'rnd':>3s} {'ms/step':>9s} {'forces':>9s} {'finish':>8s} {'MHz chip':>9s} {'MHz hwmon':>9s} {'W':>7s} {'Mcyc/step':>10s} {'G int/J':>8s}")
for r in rows:
    if "line" not in r:
        print(r)
        continue
    l, roof = r["line"], r["line"]["roofline"]
    chip = roof.get("chip") or {}
    mhz, hw, w = chip.get("delivered_mhz_by_kernel"), chip.get("sclk_mhz"), chip.get("socket_power_w")
    joule = l["value"] / w * 1e-9 if w else None
    rec = {"ms": l["ms_per_step"], "forces": roof.get("pair_forces_ms"), "finish": roof.get("pair_finish_ms"), "mhz": mhz, "hwmon": hw, "w": w, "mcyc": roof.get("mcycles_per_step"), "joule": joule}
    by.setdefault(r["ab_lib"], []).append(rec)
    f = lambda v, p=3: "-" if v is None else f"{v:.{p}f}"  # noqa: E731
    print(f"{r['ab_lib']:34s} {r['ab_round']:3d} {f(rec['ms']):>9s} {f(rec['forces']):>9s} {f(rec['finish']):>8s} {f(mhz, 0):>9s} {f(hw, 0):>9s} {f(w, 0):>7s} {f(rec['mcyc']):>10s} {f(joule, 2):>8s}")
print()
med = {}
for lib, recs in by.items():
    med[lib] = {k: statistics.median([x[k] for x in recs if x[k] is not None]) if any(x[k] is not None for x in recs) else None for k in recs[0]}
    print("median", lib, {k: (None if v is None else round(v, 4)) for k, v in med[lib].items()})
libs = list(by)
for other in libs[1:]:
    a, b = med[libs[0]], med[other]
    print(f"{other} / {libs[0]}:", {k: round(b[k] / a[k], 4) for k in a if a[k] and b[k]})
