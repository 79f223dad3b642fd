#!/bin/bash
# A/B of library variants (edge264_amd/variants/*.so) + environment settings on the bench GOP; every frame verified.  usage: tools/visits/gpu_ab2.sh TAG
TAG=${1:-ab}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp E264_ALLOW_ABLATION=1
summ() { python -c "
import json,sys
try:
    d=json.load(open(sys.argv[1])); o=d.get('other_configs') or {}
    print(sys.argv[2], d['value'], d['bit_exact'], {k.split('_')[1]:v['ms_per_launch'] for k,v in d['roofline']['kernels'].items()}, [(k[:10], v['value'], v['bit_exact']) for k,v in o.items()])
except Exception as e:
    print(sys.argv[2], 'FAILED', e)" $1 $2; }
B="python bench.py --no-cpu-baseline --no-host-packets --no-same-input --steps 8 --warmup 2 $BENCH_ARGS"
timeout 300 $B > $OUT/bench_main.json 2> $OUT/bench_main.err; summ $OUT/bench_main.json main
for lib in $(ls edge264_amd/variants/*.so 2>/dev/null); do
  n=$(basename $lib .so)
  E264_HIP_LIB=$REPO/$lib timeout 300 $B > $OUT/bench_$n.json 2> $OUT/bench_$n.err; summ $OUT/bench_$n.json $n
done
for envset in $AB_ENVS; do
  env $envset timeout 300 $B > $OUT/bench_env_$envset.json 2> $OUT/bench_env_$envset.err; summ $OUT/bench_env_$envset.json $envset
done
timeout 300 $B > $OUT/bench_main2.json 2> $OUT/bench_main2.err; summ $OUT/bench_main2.json main_again
