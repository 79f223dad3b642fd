#!/bin/bash
# round 5 A/B runner: bench of the default library, then of every variant in edge264_amd/variants with the wave counts given
# usage: tools/visits/gpu_r5b.sh TAG "variant:waves variant:waves ..."   (variant "main" = the default library)
TAG=${1:-r5b}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
summ() { python -c "
import json,sys
try:
    d=json.load(open(sys.argv[1])); o=d.get('other_configs') or {}
    print(sys.argv[2], d['value'], d['bit_exact'], {k[5:-7]:v['ms_per_launch'] for k,v in d['roofline']['kernels'].items()}, [(v['value'], v['bit_exact']) for v in o.values()])
except Exception as e:
    print(sys.argv[2], 'FAILED', e)" $1 $2; }
for vw in $2; do
  v=${vw%%:*}; w=${vw##*:}
  lib=$REPO/edge264_amd/variants/libedge264_hip_$v.so
  [ $v = main ] && lib=$REPO/edge264_amd/libedge264_hip.so
  E264_ALLOW_ABLATION=1 E264_HIP_LIB=$lib E264_WAVES=$w timeout ${BENCH_TIMEOUT:-120} python bench.py --no-cpu-baseline --no-host-packets $BENCH_ARGS > $OUT/bench_${v}_$w.json 2> $OUT/bench_${v}_$w.err; summ $OUT/bench_${v}_$w.json ${v}_$w
done
