#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3m/pairpmc
rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU_TRANS --output-format csv -d $OUT/a -- python3 $GRAFT_REPO_ROOT/tools/scratch/pair_run.py 262144 6 > $OUT/a.log 2>&1 || tail -3 $OUT/a.log
rocprofv3 --pmc SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD --output-format csv -d $OUT/b -- python3 $GRAFT_REPO_ROOT/tools/scratch/pair_run.py 262144 6 > $OUT/b.log 2>&1 || tail -3 $OUT/b.log
python3 - <<'PY'
import csv, glob, collections, os
out=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r3m/pairpmc"
for f in glob.glob(out+"/*/*/*counter_collection.csv"):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "pair_forces" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in agg.items(): print(k, sum(v)/len(v))
PY
