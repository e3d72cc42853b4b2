# rocprofv3 --kernel-trace --stats of 20 FAST pairwise steps of an in-process 8-rank world (262 144 bodies) through the REAL RCCL on one GPU:
# every kernel of every rank and RCCL's own, in one trace (round 5; run on the GPU box: gpurun -- bash tools/inprocess_world_profile.sh).
set -o pipefail
export TMPDIR=/tmp
OUT=${OUT:-gpurun_out/inprocess_world_profile}
rm -rf $OUT; mkdir -p $OUT; export OUT
python3 - <<'PY'
import sys, numpy as np
sys.path.insert(0, ".")
import __graft_entry__ as e
o = e.load_oracle().Oracle()
p, v = o.startup_state(262144, np.float32)
import os; np.savez(os.path.join(os.environ["OUT"], "in.npz"), pos=p, vel=v)
PY
export WORKER_REAL_RCCL=1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tests/fake_rccl/worker.py all $OUT/in.npz $OUT/out.npz 8 20 fast streams ws > $OUT/trace.log 2>&1 || { echo trace failed; tail -5 $OUT/trace.log; exit 1; }
find $OUT/trace -name "*kernel_stats.csv" | head -3
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
cp "$f" $OUT/kernel_stats.csv
rm -f $OUT/in.npz $OUT/out.npz
find $OUT/trace -name "*kernel_trace.csv" -size +30M -delete
head -12 $OUT/kernel_stats.csv | cut -c1-200
