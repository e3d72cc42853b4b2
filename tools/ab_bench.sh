#!/bin/bash
# tools/ab_bench.sh OUTFILE ROUNDS "ARGS of bench.py" LIB_A LIB_B ... : the headline measurement (bench.py, one JSON line per run) under several
# builds of libnbody_hip.so, interleaved ROUNDS times on ONE box (boxes differ by up to 10 % for one binary: only same-box A/B figures
# mean anything).  LIB = "-" for the product library, else a path (tools/build_pair_variant.sh makes expv/libnbody_hip_NAME.so).
# Every line is tagged {"ab_lib": ..., "ab_round": ...}; tools/ab_bench_print.py condenses the file.
out=$1; rounds=$2; args=$3; shift 3
mkdir -p "$(dirname "$out")"; : > $out
for round in $(seq 1 $rounds); do
  for lib in "$@"; do
    if [ "$lib" = "-" ]; then line=$(timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-configs $args 2>/dev/null | grep '^{' | tail -1); rc=$?
    else line=$(NBODY_HIP_LIB=$PWD/$lib timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-configs $args 2>/dev/null | grep '^{' | tail -1); rc=$?; fi
    if [ -z "$line" ]; then echo "{\"ab_lib\": \"$lib\", \"ab_round\": $round, \"error\": \"no line\"}" >> $out; exit 1; fi
    echo "{\"ab_lib\": \"$lib\", \"ab_round\": $round, \"line\": $line}" >> $out
  done
done
