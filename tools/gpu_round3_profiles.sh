#!/bin/bash
# round 3: rocprofv3 records for the headline kernel and for STRICT fp64, plus the bench lines
set -o pipefail
mkdir -p gpurun_out/r3p
PROF_OUT=gpurun_out/r3p/f32 bash tools/profile.sh --steps 20 --warmup 3 > gpurun_out/r3p/f32.log 2>&1 && echo f32 ok
PROF_OUT=gpurun_out/r3p/strict_f64 bash tools/profile.sh --steps 5 --warmup 1 --mode strict --fp64 > gpurun_out/r3p/strict_f64.log 2>&1 && echo strict_f64 ok
PROF_OUT=gpurun_out/r3p/strict_f32 bash tools/profile.sh --steps 8 --warmup 1 --mode strict > gpurun_out/r3p/strict_f32.log 2>&1 && echo strict_f32 ok
for n in 8192 16384; do
  PROF_OUT=gpurun_out/r3p/n$n bash tools/profile.sh --steps 400 --warmup 20 --bodies $n > gpurun_out/r3p/n$n.log 2>&1 && echo n$n ok
done
python3 bench.py > gpurun_out/r3p/bench.json 2> gpurun_out/r3p/bench.err && echo bench ok
