#!/bin/bash
# round-2 evidence run on the GPU box: bench lines, rehearsal of the N-rank path on one GPU, rocprofv3 profiles
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python bench.py > gpurun_out/r2_bench.txt 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/r2_bench.txt | cut -c1-600
timeout -k 10 300 python bench.py --gpus 2 --exchange host --steps 5 --no-cpu-baseline > gpurun_out/r2_bench_rehearsal2.txt 2>&1; echo "rehearsal --gpus 2 (plain shell, self-launch) rc=$?"; tail -2 gpurun_out/r2_bench_rehearsal2.txt | cut -c1-400
./tools/profile.sh > gpurun_out/r2_profile_f32.log 2>&1 && rm -rf gpurun_out/r2_prof_f32 && mv gpurun_out/prof gpurun_out/r2_prof_f32; echo "prof f32 rc=$?"
./tools/profile.sh --mode strict --steps 5 > gpurun_out/r2_profile_strict.log 2>&1 && rm -rf gpurun_out/r2_prof_strict && mv gpurun_out/prof gpurun_out/r2_prof_strict; echo "prof strict rc=$?"
./tools/profile.sh --fp64 --steps 5 > gpurun_out/r2_profile_f64.log 2>&1 && rm -rf gpurun_out/r2_prof_f64 && mv gpurun_out/prof gpurun_out/r2_prof_f64; echo "prof f64 rc=$?"
