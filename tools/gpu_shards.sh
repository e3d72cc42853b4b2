#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu.txt
timeout -k 10 300 python bench.py --no-cpu-baseline --fp64 --steps 5 > gpurun_out/bench_262144_f64.txt 2>&1; echo "rc=$?"; grep -o '"ms_per_step": [0-9.]*' gpurun_out/bench_262144_f64.txt
timeout -k 10 600 python bench.py --no-cpu-baseline --emulate-gpus 8 --sweep > gpurun_out/sweep_shard8.txt 2>&1; echo "rc=$?"; tail -1 gpurun_out/sweep_shard8.txt
timeout -k 10 600 python bench.py --no-cpu-baseline --emulate-gpus 4 --sweep > gpurun_out/sweep_shard4.txt 2>&1; echo "rc=$?"; tail -1 gpurun_out/sweep_shard4.txt
timeout -k 10 600 python bench.py --no-cpu-baseline --emulate-gpus 2 --sweep > gpurun_out/sweep_shard2.txt 2>&1; echo "rc=$?"; tail -1 gpurun_out/sweep_shard2.txt
