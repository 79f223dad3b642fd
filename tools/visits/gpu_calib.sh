#!/bin/bash
# PMC calibration (tools/calib/pmc_calib.hip) + quick check of the current build.
TAG=${1:-calib}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest_gpu.log
timeout 300 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cat $OUT/bench.json
cd /tmp
for c in RD WR; do
  if [ $c = RD ]; then CTRS="TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_EA0_RDREQ_64B TCC_EA0_RDREQ_128B"; else CTRS="TCC_EA0_WRREQ TCC_EA0_WRREQ_64B TCC_EA0_WRREQ_DRAM"; fi
  timeout 300 rocprofv3 --pmc $CTRS -d $OUT/calib_$c -- $REPO/tools/calib/pmc_calib > $OUT/calib_$c.log 2>&1; echo "calib $c rc=$?"
  timeout 400 rocprofv3 --pmc $CTRS -d $OUT/bench_$c -- python $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-verify > $OUT/bench_$c.log 2>&1; echo "bench $c rc=$?"
done
timeout 300 rocprofv3 --kernel-trace -d $OUT/calib_kt -- $REPO/tools/calib/pmc_calib > $OUT/calib_kt.log 2>&1
cd $REPO
python tools/pmc_summary.py $(find $OUT/calib_RD $OUT/calib_WR -name '*.db') > $OUT/calib_pmc.txt 2>&1; cat $OUT/calib_pmc.txt; tail -1 $OUT/calib_RD.log
python tools/rocprof_summary.py $(find $OUT/calib_kt -name '*.db' | head -1) > $OUT/calib_kt.txt 2>&1; cat $OUT/calib_kt.txt
python tools/pmc_summary.py $(find $OUT/bench_RD $OUT/bench_WR -name '*.db') --traffic $OUT/hbm_traffic.json --streams 256 --gop IPPPPPPP > $OUT/bench_pmc.txt 2>&1; cat $OUT/bench_pmc.txt
find $OUT -name '*.db' -size +20M -delete
