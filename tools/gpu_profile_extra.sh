#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
./tools/profile.sh --bodies 65536 --steps 100 > gpurun_out/profile_65536.log 2>&1 && rm -rf gpurun_out/prof_65536 && mv gpurun_out/prof gpurun_out/prof_65536; echo "prof 65536 rc=$?"
./tools/profile.sh --mode strict --steps 5 > gpurun_out/profile_strict.log 2>&1 && rm -rf gpurun_out/prof_strict && mv gpurun_out/prof gpurun_out/prof_strict; echo "prof strict rc=$?"
./tools/profile.sh --bodies 1048576 --steps 3 --warmup 1 > gpurun_out/profile_1m.log 2>&1 && rm -rf gpurun_out/prof_1m && mv gpurun_out/prof gpurun_out/prof_1m; echo "prof 1M rc=$?"
