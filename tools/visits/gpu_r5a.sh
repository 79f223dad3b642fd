#!/bin/bash
# round 5, visit 1: the matrix-pipe calibration (tools/calib/mfma_tap) + this box's baseline bench line
TAG=${1:-r5a}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 120 tools/calib/mfma_tap > $OUT/mfma_tap.txt 2>&1; echo "mfma_tap rc=$?"; cat $OUT/mfma_tap.txt
timeout 400 python bench.py --no-cpu-baseline --no-host-packets > $OUT/bench_main.json 2> $OUT/bench_main.err; echo "bench rc=$?"
python -c "
import json,sys
d=json.load(open('$OUT/bench_main.json')); o=d.get('other_configs') or {}
print('main', d['value'], d['bit_exact'], {k[5:-7]:v['ms_per_launch'] for k,v in d['roofline']['kernels'].items()}, [(v['value'], v['bit_exact']) for v in o.values()])"
