#!/bin/bash
# deblocking kernel at several waves-per-picture settings: bash tools/gpu_waves.sh TAG "7 14 ..."
TAG=${1:-wv}; REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
for w in $2; do
  timeout 300 python bench.py --no-cpu-baseline --no-other-configs --waves $w > $OUT/bench_w$w.json 2> $OUT/bench_w$w.err
  python -c "
import json,sys
d=json.load(open(sys.argv[1])); print('waves', sys.argv[2], d['value'], d['bit_exact'], {k.split('_')[1]: v['ms_per_launch'] for k, v in d['roofline']['kernels'].items()})" $OUT/bench_w$w.json $w
done
