#!/bin/bash
# same-box A/B of libnbody_hip variants under expv/: pair_run.py at 262144 f32 (and f64)
for prec in f32 f64; do
echo "base $prec: $(python3 tools/scratch/pair_run.py 262144 10 $prec)"
for lib in expv/libnbody_hip_*.so; do
  echo "$(basename $lib) $prec: $(NBODY_HIP_LIB=$PWD/$lib python3 tools/scratch/pair_run.py 262144 10 $prec)"
done
echo "base $prec again: $(python3 tools/scratch/pair_run.py 262144 10 $prec)"
done
