#!/bin/bash
# intra kernel at several waves-per-picture settings, default GOP and all-intra: bash tools/gpu_iw.sh TAG "16 8"
TAG=${1:-iw}; REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
for w in $2; do
  for gop in IPPPPPPP IIII; do
  timeout 300 python bench.py --no-cpu-baseline --no-other-configs --no-host-packets --intra-waves $w --gop $gop > $OUT/bench_w${w}_$gop.json 2> $OUT/bench_w${w}_$gop.err
  python -c "
import json,sys
d=json.load(open(sys.argv[1])); print('intra waves', sys.argv[2], sys.argv[3], d['value'], d['bit_exact'], {k.split('_')[1]: v['ms_per_launch'] for k, v in d['roofline']['kernels'].items()})" $OUT/bench_w${w}_$gop.json $w $gop
  done
done
