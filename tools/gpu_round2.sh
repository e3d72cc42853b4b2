#!/bin/bash
# round-2 evidence run on the GPU box: bench lines, rehearsal of the N-rank path on one GPU, rocprofv3 profiles
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python bench.py > gpurun_out/r2_bench.txt 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/r2_bench.txt | cut -c1-600
timeout -k 10 300 python bench.py --gpus 2 --exchange host --steps 5 --no-cpu-baseline > gpurun_out/r2_bench_rehearsal2.txt 2>&1; echo "rehearsal --gpus 2 (plain shell, self-launch) rc=$?"; tail -2 gpurun_out/r2_bench_rehearsal2.txt | cut -c1-400
./tools/profile.sh > gpurun_out/r2_profile_f32.log 2>&1 && rm -rf gpurun_out/r2_prof_f32 && mv gpurun_out/prof gpurun_out/r2_prof_f32; echo "prof f32 rc=$?"
./tools/profile.sh --mode strict --steps 5 > gpurun_out/r2_profile_strict.log 2>&1 && rm -rf gpurun_out/r2_prof_strict && mv gpurun_out/prof gpurun_out/r2_prof_strict; echo "prof strict rc=$?"
./tools/profile.sh --fp64 --steps 5 > gpurun_out/r2_profile_f64.log 2>&1 && rm -rf gpurun_out/r2_prof_f64 && mv gpurun_out/prof gpurun_out/r2_prof_f64; echo "prof f64 rc=$?"
./tools/profile.sh --bodies 65536 --steps 100 > gpurun_out/r2_profile_65536.log 2>&1 && rm -rf gpurun_out/r2_prof_65536 && mv gpurun_out/prof gpurun_out/r2_prof_65536; echo "prof 65536 rc=$?"
./tools/profile.sh --bodies 1048576 --steps 3 --warmup 1 > gpurun_out/r2_profile_1m.log 2>&1 && rm -rf gpurun_out/r2_prof_1m && mv gpurun_out/prof gpurun_out/r2_prof_1m; echo "prof 1M rc=$?"
timeout -k 10 300 python bench.py --fp64 --steps 5 > gpurun_out/r2_bench_f64.txt 2>&1; echo "bench f64 rc=$?"
timeout -k 10 300 python bench.py --mode strict --steps 5 --no-cpu-baseline > gpurun_out/r2_bench_strict.txt 2>&1; echo "bench strict rc=$?"
timeout -k 10 300 python bench.py --bodies 65536 --steps 200 --no-cpu-baseline > gpurun_out/r2_bench_65536.txt 2>&1; echo "bench 65536 rc=$?"
timeout -k 10 300 python bench.py --bodies 1048576 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r2_bench_1m.txt 2>&1; echo "bench 1M rc=$?"
for g in 2 4 8; do timeout -k 10 300 python bench.py --no-cpu-baseline --emulate-gpus $g --steps 20 > gpurun_out/r2_emulate_$g.txt 2>&1; done; cat gpurun_out/r2_emulate_*.txt | grep emulated | cut -c1-300
{ for n in 1024 4096; do ./cuda-nbody_amd/nbody --benchmark --numbodies=$n -i 2000 | grep "bodies, total\|billion"; ./cuda-nbody_amd/nbody --benchmark --numbodies=$n -i 2000 --graph | grep "bodies, total\|billion"; done; } > gpurun_out/r2_smalln_cli.txt 2>&1; cat gpurun_out/r2_smalln_cli.txt
./cuda-nbody_amd/nbody --benchmark --numbodies=262144 > gpurun_out/r2_cli_bench.txt 2>&1; tail -3 gpurun_out/r2_cli_bench.txt
