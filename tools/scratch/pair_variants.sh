#!/bin/bash
n=${1:-262144}; prec=${2:-f32}
echo "base: $(python3 tools/scratch/pair_run.py $n 10 $prec)"
for lib in expv/libnbody_hip_*.so; do
  echo "$(basename $lib): $(NBODY_HIP_LIB=$PWD/$lib python3 tools/scratch/pair_run.py $n 10 $prec)"
done
echo "base again: $(python3 tools/scratch/pair_run.py $n 10 $prec)"
