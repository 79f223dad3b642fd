#!/bin/bash
# One GPU-box visit: parity tests, default bench, rocprofv3 kernel trace of the same command, HBM PMC passes.
# Usage (from the repo root on the GPU box):  bash tools/gpu_round.sh [tag]
TAG=${1:-r01}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -3 $OUT/pytest_gpu.log
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
cat $OUT/bench_default.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_kt -- python $REPO/bench.py --no-cpu-baseline --no-verify > $OUT/bench_kt.json 2> $OUT/bench_kt.err; echo "kt rc=$?"
timeout 600 rocprofv3 --pmc TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_EA0_RDREQ_64B TCC_EA0_RDREQ_128B -d $OUT/pmc_fetch -- python $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-verify > $OUT/bench_pmc_fetch.json 2> $OUT/bench_pmc_fetch.err; echo "fetch rc=$?"
timeout 600 rocprofv3 --pmc TCC_EA0_WRREQ TCC_EA0_WRREQ_64B -d $OUT/pmc_write -- python $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-verify > $OUT/bench_pmc_write.json 2> $OUT/bench_pmc_write.err; echo "write rc=$?"
cd $REPO
KT=$(find $OUT/prof_kt -name '*.db' | head -1)
FE=$(find $OUT/pmc_fetch -name '*.db' | head -1)
WR=$(find $OUT/pmc_write -name '*.db' | head -1)
python tools/rocprof_summary.py $KT > $OUT/kernel_stats.txt 2>&1; cat $OUT/kernel_stats.txt
find $OUT/prof_kt -name '*stats*.csv' | head; 
python tools/pmc_summary.py $FE $WR --traffic $OUT/hbm_traffic.json --streams 256 --gop IPPPPPPP > $OUT/pmc_summary.txt 2>&1; grep -v 'LEVEL\|_DRAM' $OUT/pmc_summary.txt
du -sh $OUT
# keep the merge small: databases are summarised above
find $OUT -name '*.db' -size +20M -delete
