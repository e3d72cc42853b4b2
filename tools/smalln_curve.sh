#!/bin/bash
# step rate of the CLI benchmark for small systems (launch- and occupancy-bound range)
for n in 1024 2048 4096 8192 16384 32768 65536; do
  it=$(( 20000000 / n )); [ $it -gt 4000 ] && it=4000; [ $it -lt 50 ] && it=50
  ./cuda-nbody_amd/nbody --benchmark --numbodies=$n -i $it | grep "bodies, total\|billion" | tr '\n' ' '; echo
done
