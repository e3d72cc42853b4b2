#!/bin/bash
# pair_finish / pair_forces kernel times at a few sizes (rocprofv3 kernel trace)
export TMPDIR=/tmp
for n in 16384 65536 262144 1048576; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/r3i/t$n; rm -rf $OUT; mkdir -p $OUT
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/scratch/pair_run.py $n 20 > $OUT.log 2>&1)
  echo "n=$n"; grep pair_ $OUT/*/*kernel_stats.csv | cut -d, -f1-4 | sed 's/.*pair_/pair_/' 
done
