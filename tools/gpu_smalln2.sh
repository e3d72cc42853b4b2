#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 120 ./tools/grid_barrier_probe > gpurun_out/r2_grid_barrier_probe.txt 2>&1; echo rc=$?; cat gpurun_out/r2_grid_barrier_probe.txt
for n in 1024 4096 16384; do timeout -k 10 100 python3 tools/plan_probe.py --bodies $n --steps 2000 0,0,0; done > gpurun_out/r2_smalln.txt 2>&1; cat gpurun_out/r2_smalln.txt
