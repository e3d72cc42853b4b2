#!/usr/bin/env python3
"""print a tools/ab.sh output file compactly"""
import json
import sys

for l in open(sys.argv[1]):
    if l.startswith("=="):
        print(l.strip())
        continue
    try:
        r = json.loads(l)
    except ValueError:
        print(l.rstrip()[:200])
        continue
    print(r["bodies"], r["one_sided_us"], [(tuple(p["plan"]), p["us"], p["forces_finish_us"]) for p in r["plans"]])
