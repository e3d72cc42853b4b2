#!/bin/bash
for n in 1024 4096 16384 65536; do for g in "" "--graph"; do echo "== N=$n $g"; cuda-nbody_amd/nbody --benchmark --numbodies=$n -i 1000 $g 2>&1 | grep -E "total time|billion"; done; done
