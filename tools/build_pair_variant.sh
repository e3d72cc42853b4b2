#!/bin/bash
# tools/build_pair_variant.sh NAME "FLAGS": libnbody_hip.so with nbody_pair.hip recompiled under FLAGS -> expv/libnbody_hip_NAME.so
# (A/B runs on one box: NBODY_HIP_LIB=expv/libnbody_hip_NAME.so; expv/ travels to the GPU box, it is git-ignored)
set -e
cd "$(dirname "$0")/.."
mkdir -p expv
S=cuda-nbody_amd/csrc
# the other objects must be CURRENT: PairArgs / FinishArgs cross object boundaries by value, and a variant linked against objects
# built before a header change mixes struct layouts (round 4's GPU memory fault was exactly that, in the main build)
make -s -j8 -C $S
/opt/rocm/bin/hipcc -O3 -std=c++20 --offload-arch=gfx950 -fPIC -fvisibility=hidden -w $2 -c $S/nbody_pair.hip -o expv/pair_$1.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o expv/libnbody_hip_$1.so $S/nbody_strict.o $S/nbody_fast.o expv/pair_$1.o $S/nbody_capi.o $S/nbody_comm.o -ldl -lpthread
rm -f expv/pair_$1.o
