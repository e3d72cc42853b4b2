#!/bin/bash
# A/B of hipcc scheduler strategies for the fast kernels, interleaved on one box (rule 24: same process family, same device)
for round in 1 2; do
  for v in "" _max-ilp _max-memory-clause _iterative-ilp _iterative-minreg; do
    lib=$PWD/cuda-nbody_amd/libnbody_hip$v.so
    r=$(NBODY_HIP_LIB=$lib python bench.py --no-cpu-baseline --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms  %.1f G/s' % (d['ms_per_step'], d['value']*1e-9))")
    r64=$(NBODY_HIP_LIB=$lib python bench.py --no-cpu-baseline --fp64 --steps 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms' % d['ms_per_step'])")
    echo "round $round variant '${v:-default}': fp32 $r | fp64 $r64"
  done
done
