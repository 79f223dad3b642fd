#!/bin/bash
# bench sweeps given as lines of extra arguments in $2.. (quoted strings)
TAG=$1; shift
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
i=0
for a in "$@"; do
  i=$((i+1))
  timeout 300 python bench.py --no-cpu-baseline --no-verify $a > $OUT/b$i.json 2> $OUT/b$i.err
  python -c "
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[2], '|', d['value'], d['roofline']['kernel_ms_per_launch'])" $OUT/b$i.json "$a"
done
