#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu.txt
./tools/profile.sh > gpurun_out/profile_f32.log 2>&1 && mv gpurun_out/prof gpurun_out/prof_f32; echo "prof f32 rc=$?"
./tools/profile.sh --fp64 --steps 5 > gpurun_out/profile_f64.log 2>&1 && mv gpurun_out/prof gpurun_out/prof_f64; echo "prof f64 rc=$?"
timeout -k 10 300 python bench.py > gpurun_out/bench.txt 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/bench.txt | cut -c1-400
timeout -k 10 300 python bench.py --fp64 --steps 5 > gpurun_out/bench_f64.txt 2>&1; echo "bench f64 rc=$?"; tail -1 gpurun_out/bench_f64.txt | cut -c1-300
