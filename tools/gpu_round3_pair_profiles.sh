#!/bin/bash
# round 3: rocprofv3 records of the pairwise layout (the headline since round 3) -- kernel trace + the separate PMC passes
set -o pipefail
mkdir -p gpurun_out/r3q
PROF_OUT=gpurun_out/r3q/f32 bash tools/profile.sh --steps 20 --warmup 3 > gpurun_out/r3q/f32.log 2>&1 && echo f32 ok
PROF_OUT=gpurun_out/r3q/f64 bash tools/profile.sh --steps 8 --warmup 2 --fp64 > gpurun_out/r3q/f64.log 2>&1 && echo f64 ok
PROF_OUT=gpurun_out/r3q/n65536 bash tools/profile.sh --steps 200 --warmup 10 --bodies 65536 > gpurun_out/r3q/n65536.log 2>&1 && echo n65536 ok
PROF_OUT=gpurun_out/r3q/n16384 bash tools/profile.sh --steps 400 --warmup 20 --bodies 16384 > gpurun_out/r3q/n16384.log 2>&1 && echo n16384 ok
PROF_OUT=gpurun_out/r3q/n1m bash tools/profile.sh --steps 4 --warmup 1 --bodies 1048576 > gpurun_out/r3q/n1m.log 2>&1 && echo n1m ok
python3 bench.py > gpurun_out/r3q/bench.json 2> gpurun_out/r3q/bench.err && echo bench ok
python3 bench.py --layout one-sided --no-configs > gpurun_out/r3q/bench_one_sided.json 2> gpurun_out/r3q/bench_one_sided.err && echo bench one-sided ok
