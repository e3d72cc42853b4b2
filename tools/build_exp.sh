#!/bin/bash
# tools/build_exp.sh NAME "FLAGS" -- builds an experimental/diagnostic variant of libnbody_hip.so into exp/libnbody_hip_NAME.so
# (never the product library; select it with NBODY_HIP_LIB=exp/libnbody_hip_NAME.so).
set -e
cd "$(dirname "$0")/.."
mkdir -p exp
C="/opt/rocm/bin/hipcc -O3 -std=c++20 --offload-arch=gfx950 -fPIC -fvisibility=hidden -w $2"
S=cuda-nbody_amd/csrc
$C -ffp-contract=off -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=max-ilp -c $S/nbody_strict.hip -o exp/strict_$1.o &
$C -c $S/nbody_fast.hip -o exp/fast_$1.o &
$C -c $S/nbody_capi.hip -o exp/capi_$1.o &
$C -c $S/nbody_comm.hip -o exp/comm_$1.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o exp/libnbody_hip_$1.so exp/strict_$1.o exp/fast_$1.o exp/capi_$1.o exp/comm_$1.o -ldl
rm -f exp/*_$1.o
ls -la exp/libnbody_hip_$1.so
