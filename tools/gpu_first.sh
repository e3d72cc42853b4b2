#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.txt 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/pytest_gpu.txt
timeout -k 10 300 python bench.py --sweep > gpurun_out/sweep.txt 2>&1; echo "sweep rc=$?"
tail -3 gpurun_out/sweep.txt
timeout -k 10 300 python bench.py > gpurun_out/bench.txt 2>&1; echo "bench rc=$?"
tail -2 gpurun_out/bench.txt
