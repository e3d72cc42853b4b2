#!/bin/bash
# rocprofv3 evidence for the dominant kernel (run on the GPU box through gpurun).
# Counters go in their own passes, never combined with any trace option other than --kernel-trace.
set -o pipefail
export TMPDIR=/tmp
OUT=${PROF_OUT:-gpurun_out/prof}
rm -rf $OUT; mkdir -p $OUT
ARGS="bench.py --no-cpu-baseline --no-configs ${*:---steps 20 --warmup 3}"  # --no-configs: only the headline kernel in the trace
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.log 2>&1 || { echo trace failed; tail -5 $OUT/trace.log; exit 1; }
echo trace ok
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ARGS > $OUT/pmc_fetch.log 2>&1 || { echo fetch failed; tail -5 $OUT/pmc_fetch.log; exit 1; }
echo fetch ok
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ARGS > $OUT/pmc_write.log 2>&1 || { echo write failed; tail -5 $OUT/pmc_write.log; exit 1; }
echo write ok
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq -- python3 $ARGS > $OUT/pmc_sq.log 2>&1 || { echo sq failed; tail -5 $OUT/pmc_sq.log; }
echo sq done
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_TRANS --output-format csv -d $OUT/pmc_sq2 -- python3 $ARGS > $OUT/pmc_sq2.log 2>&1 || { echo sq2 failed; tail -5 $OUT/pmc_sq2.log; }
echo sq2 done
find $OUT -name "*.csv" | head -40
