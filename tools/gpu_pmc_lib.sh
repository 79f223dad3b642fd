#!/bin/bash
# HBM read / write request counters of one library variant: bash tools/gpu_pmc_lib.sh TAG LIB.so
TAG=$1; LIB=$2; REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp E264_HIP_LIB=$REPO/$LIB
B="python $REPO/bench.py --no-cpu-baseline --no-verify --no-other-configs --no-host-packets --steps 1 --warmup 0"
cd /tmp
timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_EA0_RDREQ_64B TCC_EA0_RDREQ_128B -d $OUT/rd -- $B > /dev/null 2>&1
timeout 300 rocprofv3 --pmc TCC_EA0_WRREQ TCC_EA0_WRREQ_64B -d $OUT/wr -- $B > /dev/null 2>&1
cd $REPO
python tools/pmc_summary.py $(find $OUT/rd $OUT/wr -name '*.db') > $OUT/requests.txt 2>&1
grep -A6 deblock $OUT/requests.txt
find $OUT -name '*.db' -delete
