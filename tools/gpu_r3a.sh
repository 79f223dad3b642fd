#!/bin/bash
# Round 3, visit A: the refactored back end (lanes, upload queue, recycler) -- parity, lanes sweep, staging microbenchmark.
TAG=${1:-r3a}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
summ() { python -c "
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    p=d.get('pcie_inclusive') or {}
    print(sys.argv[2], d['value'], 'exact', d['bit_exact'], d['roofline']['kernels'] and {k[5:-7]:v['ms_per_launch'] for k,v in d['roofline']['kernels'].items()}, 'e2e', d['roofline']['end_to_end']['ms_per_submission'], 'pcie', p.get('value'), (p.get('pinned_in_place') or {}).get('value'))
except Exception as e:
    print(sys.argv[2], 'FAILED', e)" $1 $2; }
run() { n=$1; shift; timeout 300 python bench.py --no-cpu-baseline "$@" > $OUT/bench_$n.json 2> $OUT/bench_$n.err; summ $OUT/bench_$n.json $n; }
run l1 --no-other-configs
run l2 --lanes 2 --no-other-configs
E264_UPLOAD_QUEUE=0 run l1_noup --no-other-configs --no-verify
Q="--no-other-configs --no-verify --no-host-packets"
run l4 --lanes 4 $Q
run l2_w4 --lanes 2 --waves 4 $Q
run l2_w4_i8 --lanes 2 --waves 4 --intra-waves 8 $Q
run l4_w4_i8 --lanes 4 --waves 4 --intra-waves 8 $Q
run l1_w4_i8 --lanes 1 --waves 4 --intra-waves 8 $Q
run s512_l1 --streams 512 --lanes 1 $Q
run s512_l2 --streams 512 --lanes 2 $Q
run s512_l4 --streams 512 --lanes 4 $Q
run s512_l4_w4_i8 --streams 512 --lanes 4 --waves 4 --intra-waves 8 $Q
run s1024_l4_w4_i8 --streams 1024 --lanes 4 --waves 4 --intra-waves 8 $Q
timeout 120 tools/calib/stage_rate > $OUT/stage_rate.txt 2>&1; cat $OUT/stage_rate.txt
timeout 120 tools/calib/load_rate > $OUT/load_rate.txt 2>&1; cat $OUT/load_rate.txt
