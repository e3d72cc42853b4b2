#!/bin/bash
for cfg in "0 0 0" "4 8 4" "4 8 16" "4 16 4" "4 16 8" "2 8 8" "2 8 4" "4 4 16"; do
  python3 tools/pair_rank_probe.py 262144 f32 $cfg | grep -E "override|ranks\": 8|ranks\": 4" | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['ranks'], [r['pairwise_ms'] for r in d['per_rank']], d['projected_speedup_vs_single_gpu_pairwise_excluding_exchange'])
    else: print(l.strip())
"
done
