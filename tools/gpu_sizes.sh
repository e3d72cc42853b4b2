#!/bin/bash
# secondary configurations + projections (one GPU)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu.txt
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/bench_262144.txt 2>&1; echo "rc=$?"
timeout -k 10 300 python bench.py --no-cpu-baseline --bodies 65536 --steps 100 > gpurun_out/bench_65536.txt 2>&1; echo "rc=$?"
timeout -k 10 300 python bench.py --no-cpu-baseline --bodies 65536 --sweep > gpurun_out/sweep_65536.txt 2>&1; echo "rc=$?"
timeout -k 10 300 python bench.py --no-cpu-baseline --fp64 --steps 5 > gpurun_out/bench_262144_f64.txt 2>&1; echo "rc=$?"
timeout -k 10 600 python bench.py --no-cpu-baseline --fp64 --sweep > gpurun_out/sweep_262144_f64.txt 2>&1; echo "rc=$?"
timeout -k 10 300 python bench.py --no-cpu-baseline --bodies 1048576 --steps 3 --warmup 1 > gpurun_out/bench_1M.txt 2>&1; echo "rc=$?"
timeout -k 10 300 python bench.py --no-cpu-baseline --mode strict --steps 3 --warmup 1 > gpurun_out/bench_strict.txt 2>&1; echo "rc=$?"
for g in 2 4 8; do timeout -k 10 300 python bench.py --no-cpu-baseline --emulate-gpus $g --steps 20 > gpurun_out/emulate_$g.txt 2>&1; echo "emulate $g rc=$?"; done
timeout -k 10 120 cuda-nbody_amd/nbody --benchmark --numbodies=262144 -i 10 > gpurun_out/cli_bench.txt 2>&1; echo "cli rc=$?"
timeout -k 10 120 cuda-nbody_amd/nbody --benchmark --numbodies=65536 -i 100 >> gpurun_out/cli_bench.txt 2>&1; echo "cli rc=$?"
timeout -k 10 120 cuda-nbody_amd/nbody --benchmark --numbodies=262144 --fp64 -i 3 >> gpurun_out/cli_bench.txt 2>&1; echo "cli rc=$?"
timeout -k 10 120 cuda-nbody_amd/nbody --benchmark --numbodies=16384 --hostmem -i 20 >> gpurun_out/cli_bench.txt 2>&1; echo "cli rc=$?"
grep -h '"value"' gpurun_out/bench_*.txt | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print(d['config']['bodies'], d['dtype'], d['config']['workload'][-12:], 'ms/step %.3f'%d['ms_per_step'], 'G inter/s %.1f'%(d['value']*1e-9), 'frac %.3f'%d['roofline']['frac'], d['config']['kernel_plan'])
"
cat gpurun_out/emulate_*.txt | grep emulated
grep -E "bodies, total|billion|GFLOP" gpurun_out/cli_bench.txt
