#!/bin/bash
# tools/ab.sh OUTFILE "ARGS of tools/pair_plan_times.py" LIB_A LIB_B ... : the same timing under several builds of libnbody_hip.so, interleaved
# twice on ONE box (boxes differ by up to 10 %, so only same-box A/B figures mean anything).  LIB = "-" for the product library.
out=$1; args=$2; shift 2
mkdir -p "$(dirname "$out")"; : > $out
for round in 1 2; do
  for lib in "$@"; do
    echo "== round $round lib $lib" >> $out
    if [ "$lib" = "-" ]; then timeout -k 10 300 python tools/pair_plan_times.py $args >> $out 2>&1
    else NBODY_HIP_LIB=$PWD/$lib timeout -k 10 300 python tools/pair_plan_times.py $args >> $out 2>&1; fi
  done
done
