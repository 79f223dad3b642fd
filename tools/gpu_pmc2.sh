#!/bin/bash
# PMC passes over the default bench command: bash tools/gpu_pmc2.sh TAG "CTR1 CTR2 ..." ["CTRS of pass 2" ...]
TAG=$1; shift
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for set in "$@"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set -d $OUT/p$i -- python $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-verify > $OUT/p$i.log 2>&1; echo "pass $i rc=$?"
done
cd $REPO
python tools/pmc_summary.py $(find $OUT -name '*.db') > $OUT/summary.txt 2>&1; grep -A40 "pred_kernel" $OUT/summary.txt | head -60
find $OUT -name '*.db' -size +20M -delete
