#!/bin/bash
# does an even split of the units over the waves pay?  (n_units = (NB/2+1)*I; waves = C*S)
run() { python3 tools/scratch/pair_run.py $1 300 f32 $2 $3 $4 | sed "s/^/R=$2 S=$3 C=$4: /"; }
echo "16384 (R=2: 132 units per block)"; run 16384 2 8 4; run 16384 2 4 11; run 16384 2 4 3; run 16384 2 4 8; run 16384 2 4 16; run 16384 2 8 2
echo "32768 (R=4: 264 units)"; run 32768 4 8 4; run 32768 4 8 3; run 32768 4 8 11; run 32768 4 4 11; run 32768 4 4 6
echo "65536 (R=4: 520 units)"; run 65536 4 8 4; run 65536 4 8 5; run 65536 4 8 13; run 65536 4 4 10; run 65536 4 4 13
echo "131072 (R=4: 1032 units)"; run 131072 4 8 4; run 131072 4 8 3; run 131072 4 8 1; run 131072 4 4 6
