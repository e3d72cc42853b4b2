#!/bin/bash
# geometry sweep of the pairwise layout: R S C
n=${1:-262144}; prec=${2:-f32}
for cfg in "4 8 1" "4 16 1" "4 8 2" "4 8 4" "4 16 2" "4 4 1" "4 4 2" "2 8 1" "2 16 1" "2 8 2"; do
  python3 tools/scratch/pair_run.py $n 10 $prec $cfg
done
