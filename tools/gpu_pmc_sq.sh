#!/bin/bash
# SQ instruction-mix / wait counters of the bench kernels (one pass, 8 SQ counters) + counter list
TAG=${1:-sq}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L > $OUT/counters.txt 2>&1
grep -o "TCC_EA[A-Z0-9_]*" $OUT/counters.txt | sort -u | tr '\n' ' '; echo
timeout 400 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM -d $OUT/sq1 -- python $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-verify > $OUT/sq1.log 2>&1; echo "sq1 rc=$?"
timeout 400 rocprofv3 --pmc SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES -d $OUT/sq2 -- python $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-verify > $OUT/sq2.log 2>&1; echo "sq2 rc=$?"
cd $REPO
python tools/pmc_summary.py $(find $OUT/sq1 $OUT/sq2 -name '*.db') > $OUT/sq_summary.txt 2>&1; cat $OUT/sq_summary.txt
find $OUT -name '*.db' -size +20M -delete
