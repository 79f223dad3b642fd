#!/bin/bash
# SQ counters of e264_intra_kernel on all-intra pictures (bench --gop II: the I frame mix of the bench GOP), for the default
# library and every variant in edge264_amd/variants: instruction mix, issue and wait cycles, LDS conflicts.
# usage: bash tools/visits/gpu_pmc_intra.sh TAG
TAG=${1:-pmci}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
B="python $REPO/bench.py --gop II --steps 1 --warmup 0 --no-cpu-baseline --no-verify --no-other-configs --no-host-packets"
cd /tmp
for lib in main $(ls $REPO/edge264_amd/variants/*.so 2>/dev/null); do
  n=$(basename $lib .so); env=""
  [ "$lib" != main ] && export E264_HIP_LIB=$lib || unset E264_HIP_LIB
  timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_INST_ANY -d $OUT/a_$n -- $B > /dev/null 2> $OUT/a_$n.err
  timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY -d $OUT/b_$n -- $B > /dev/null 2> $OUT/b_$n.err
  python $REPO/tools/pmc_summary.py $(find $OUT/a_$n $OUT/b_$n -name '*.db') > $OUT/pmc_$n.txt 2>&1
  echo "== $n"; grep -A20 "intra_kernel" $OUT/pmc_$n.txt | grep -v "^_Z" | head -20
done
find $OUT -name '*.db' -size +5M -delete
