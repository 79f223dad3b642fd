#!/bin/bash
# streams x waves sweep: bash tools/gpu_sweep2.sh TAG "streams:waves:intra_waves ..."
TAG=$1; shift
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
for spec in $1; do
  IFS=: read s w iw <<< "$spec"
  timeout 300 python bench.py --no-cpu-baseline --no-verify --no-other-configs --no-host-packets --steps 3 --streams $s --waves $w --intra-waves $iw > $OUT/b_$s_$w_$iw.json 2> $OUT/err.txt
  python -c "
import json,sys
d=json.load(open(sys.argv[1])); k=d['roofline']['kernels']; print(sys.argv[2], d['value'], {n.split('_')[1]: v['ms_per_launch'] for n,v in k.items()})" $OUT/b_$s_$w_$iw.json $spec
done
