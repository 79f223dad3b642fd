#!/bin/bash
# Round 4, visit A: A/B of prediction-kernel variants (edge264_amd/variants/*.so) on the bench GOP, every frame verified.
TAG=${1:-r4a}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
summ() { python -c "
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    print(sys.argv[2], d['value'], d['bit_exact'], {k.split('_')[1]:v['ms_per_launch'] for k,v in d['roofline']['kernels'].items()})
except Exception as e:
    print(sys.argv[2], 'FAILED', e)" $1 $2; }
B="python bench.py --no-cpu-baseline --no-host-packets --no-other-configs --steps 8 --warmup 2"
timeout 300 $B > $OUT/bench_main.json 2> $OUT/bench_main.err; summ $OUT/bench_main.json main
for lib in $(ls edge264_amd/variants/*.so 2>/dev/null); do
  n=$(basename $lib .so)
  E264_HIP_LIB=$REPO/$lib timeout 300 $B > $OUT/bench_$n.json 2> $OUT/bench_$n.err; summ $OUT/bench_$n.json $n
done
timeout 300 $B > $OUT/bench_main2.json 2> $OUT/bench_main2.err; summ $OUT/bench_main2.json main_again
