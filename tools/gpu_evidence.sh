#!/bin/bash
# tools/gpu_evidence.sh [OUT] -- the evidence run of a round, sent to the GPU box as ONE gpurun call:
#     gpurun --timeout 1100 -- 'bash tools/gpu_evidence.sh gpurun_out/r4'
# (1) the GPU test suite, (2) smoke, (3) the default bench line and the one-sided one, (3b, round 5) the real-RCCL self-loop and the loopback
# contention measurement, (4) rocprofv3 kernel trace + the separate
# PMC passes (tools/profile.sh) of every BASELINE size through the bench command itself.  Afterwards, here:
#     python tools/summarize_prof.py gpurun_out/r4/f32 profiles/round4_n262144_f32_pairwise pair_forces      (and so on per size;
#     one_sided -> profiles/roundN_n262144_f32 integrate_bodies, strict -> ..._strict integrate_bodies_strict)
# Steps are joined so that a GPU step that fails or times out stops the run (never start GPU work after a timeout).
set -o pipefail
OUT=${1:-gpurun_out/evidence}
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1; rc=$?; tail -3 $OUT/pytest_gpu.txt; [ $rc -eq 0 ] || exit $rc
timeout -k 10 120 python __graft_entry__.py smoke > $OUT/smoke.txt 2>&1 || { tail -5 $OUT/smoke.txt; exit 1; }
timeout -k 10 300 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
echo "bench ok: $(cut -c1-200 $OUT/bench.json)"
timeout -k 10 300 python3 bench.py --layout one-sided --no-configs --no-cpu-baseline > $OUT/bench_one_sided.json 2> $OUT/bench_one_sided.err || exit 1
# round 5: the real RCCL on this one GPU -- the self-loop (every byte checked) and the loopback rank next to the shipping kernels
for how in "" "--torch"; do timeout -k 10 150 python3 tools/rccl_selfloop.py $how 2>/dev/null | grep "^{" >> $OUT/rccl_selfloop.jsonl || { echo "rccl_selfloop $how FAILED"; exit 1; }; done
for a in "--world 5 --slice 4096 --steps 3" "--world 7 --slice 32768 --steps 2 --torch"; do timeout -k 10 250 python3 tools/rccl_loopback_parity.py $a 2>/dev/null | grep "^{" >> $OUT/rccl_loopback_parity.jsonl || { echo "rccl_loopback_parity $a FAILED"; exit 1; }; done
for w in 8 4 2; do timeout -k 10 300 python3 tools/exchange_contention.py --bodies 262144 --world $w 2>/dev/null | grep "^{" >> $OUT/exchange_contention.jsonl || { echo "exchange_contention $w FAILED"; exit 1; }; done
run_prof() { PROF_OUT=$OUT/$1 timeout -k 10 600 bash tools/profile.sh "${@:2}" > $OUT/$1.log 2>&1 && echo "$1 ok" || { echo "$1 FAILED"; tail -5 $OUT/$1.log; return 1; }; }
run_prof f32 --steps 20 --warmup 3 &&
run_prof n65536 --steps 200 --warmup 10 --bodies 65536 &&
run_prof n16384 --steps 400 --warmup 20 --bodies 16384 &&
run_prof f64 --steps 8 --warmup 2 --fp64 &&
run_prof n1m --steps 4 --warmup 1 --bodies 1048576 &&
run_prof strict --steps 8 --warmup 2 --mode strict &&
run_prof n50000 --steps 200 --warmup 10 --bodies 50000 &&
run_prof one_sided --steps 20 --warmup 3 --layout one-sided
