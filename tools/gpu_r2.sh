#!/bin/bash
# round-2 GPU visit: parity tests, bench of the default build and of every
# variant library; optional rocprofv3 kernel trace.   bash tools/gpu_r2.sh TAG [prof]
TAG=${1:-r2}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
if [ -z "$NO_PYTEST" ]; then
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_gpu.log
fi
summ() { python -c "
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[2], d['value'], d['bit_exact'], {k.split('_')[1]: v['ms_per_launch'] for k, v in d['roofline']['kernels'].items()})" $1 $2; }
timeout 300 python bench.py --no-cpu-baseline > $OUT/bench_main.json 2> $OUT/bench_main.err; summ $OUT/bench_main.json main
for lib in edge264_amd/variants/*.so; do
  [ -f "$lib" ] || continue
  n=$(basename $lib .so)
  E264_HIP_LIB=$REPO/$lib timeout 300 python bench.py --no-cpu-baseline > $OUT/bench_$n.json 2> $OUT/bench_$n.err; summ $OUT/bench_$n.json $n
done
if [ -n "$2" ]; then
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/prof_kt -- python $REPO/bench.py --no-cpu-baseline --no-verify > $OUT/bench_kt.json 2> $OUT/bench_kt.err; echo "kt rc=$?"
timeout 400 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM -d $OUT/sq1 -- python $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-verify > $OUT/sq1.log 2>&1
cd $REPO
python tools/rocprof_summary.py $(find $OUT/prof_kt -name '*.db' | head -1) > $OUT/kernel_stats.txt 2>&1; cat $OUT/kernel_stats.txt
python tools/pmc_summary.py $(find $OUT/sq1 -name '*.db') > $OUT/sq_summary.txt 2>&1; cat $OUT/sq_summary.txt
find $OUT -name '*.db' -size +20M -delete
fi
