#!/usr/bin/env python3
"""oracle/cpu_benchmark.py -- TEST INFRASTRUCTURE: what `cuda-nbody --cpu --benchmark --numbodies=N [-i K] [--fp64]`
does, on the CPU restatement in this directory: start-up bodies from the third rand() segment,
K x BodySystemCPU<T>::update(0.016) between two steady-clock reads, no warm-up (compute_cpu.cpp:72-80), and the
reference's three result lines (compute.cpp:105-121).  BASELINE.json configs[0] is `--numbodies 1024 -i 100`.

    python oracle/cpu_benchmark.py --numbodies 1024 -i 100 [--fp64] [--openmp]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import oracle as O  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--numbodies", type=int, default=4096)  # ComputeCPU's default, compute_cpu.cpp:31
    ap.add_argument("-i", "--iterations", type=int, default=10)
    ap.add_argument("--fp64", action="store_true")
    ap.add_argument("--openmp", action="store_true")
    args = ap.parse_args()
    dtype = np.float64 if args.fp64 else np.float32
    if not args.fp64 and args.numbodies % 8:
        sys.exit("the reference's fp32 CPU path needs a multiple of 8 bodies (bodysystemcpu.cpp:168)")
    orc = O.Oracle(openmp=args.openmp)
    print("> Simulation with CPU" + (" using OpenMP" if args.openmp else ""))
    print("> Simulation data stored in system memory")
    print(f"> {'Double' if args.fp64 else 'Single'} precision floating point simulation")
    print("> 0 Devices used for simulation")
    print(f"number of bodies = {args.numbodies}")
    pos, vel = orc.startup_state(args.numbodies, dtype)
    ms = np.float32(orc.benchmark(pos, vel, O.DEMO0["time_step"], args.iterations))
    n = args.numbodies
    flops = 30 if args.fp64 else 20
    inter = np.float32(n * n) * np.float32(1e-9) * (np.float32(args.iterations) * (np.float32(1000.0) / ms))
    print(f"{n} bodies, total time for {args.iterations} iterations: {ms} ms")
    print(f"= {inter} billion interactions per second")
    print(f"= {inter * np.float32(flops)} {'double' if args.fp64 else 'single'}-precision GFLOP/s at {flops} flops per interaction")
    print(f"({orc.num_threads() if args.openmp else 1} thread(s))")


if __name__ == "__main__":
    main()
