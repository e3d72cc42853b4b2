#!/bin/bash
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3m/mbpmc
rm -rf $OUT; mkdir -p $OUT; cd /tmp
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d $OUT/a -- $GRAFT_REPO_ROOT/tools/sym_microbench quick > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- $GRAFT_REPO_ROOT/tools/sym_microbench quick > $OUT/t.log 2>&1
grep cycles $OUT/a.log $OUT/t.log
python3 - <<'PY'
import csv, glob, collections, os
out=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r3m/mbpmc"
for f in glob.glob(out+"/a/*/*counter_collection.csv"):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in agg.items(): print(k, {c: sum(x)/len(x) for c,x in v.items()})
for f in glob.glob(out+"/t/*/*kernel_stats.csv"): print(open(f).read())
PY
