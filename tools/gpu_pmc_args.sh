#!/bin/bash
# SQ instruction mix for bench runs with extra args: gpu_pmc_args.sh TAG 'args1' 'args2' ...
TAG=$1; shift
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for a in "$@"; do
  i=$((i+1))
  timeout 400 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM -d $OUT/sq$i -- python $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-verify $a > $OUT/sq$i.log 2>&1
  echo "=== $a"
  python $REPO/tools/pmc_summary.py $(find $OUT/sq$i -name '*.db') | grep -A8 "mbpar\|deblock_kernel" | grep -v "^--"
done
find $OUT -name '*.db' -size +20M -delete
