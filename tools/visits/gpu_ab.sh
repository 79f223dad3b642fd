#!/bin/bash
# A/B of library variants: parity tests on the default build, then bench per variant (edge264_amd/variants/*.so)
# usage: tools/visits/gpu_ab.sh TAG [pmc] ; extra bench arguments through $BENCH_ARGS
TAG=${1:-ab}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest_gpu.log
summ() { python -c "
import json,sys
try:
    d=json.load(open(sys.argv[1])); o=d.get('other_configs') or {}
    print(sys.argv[2], d['value'], d['bit_exact'], {k[5:-7]:v['ms_per_launch'] for k,v in d['roofline']['kernels'].items()}, [(v['value'], v['bit_exact']) for v in o.values()])
except Exception as e:
    print(sys.argv[2], 'FAILED', e)" $1 $2; }
timeout 300 python bench.py --no-cpu-baseline --no-host-packets $BENCH_ARGS > $OUT/bench_main.json 2> $OUT/bench_main.err; summ $OUT/bench_main.json main
for lib in $(ls edge264_amd/variants/*.so 2>/dev/null); do
  n=$(basename $lib .so)
  E264_HIP_LIB=$REPO/$lib timeout 300 python bench.py --no-cpu-baseline --no-host-packets $BENCH_ARGS > $OUT/bench_$n.json 2> $OUT/bench_$n.err; summ $OUT/bench_$n.json $n
done
if [ -n "$2" ]; then
cd /tmp
timeout 400 rocprofv3 --pmc TCC_EA0_WRREQ TCC_EA0_WRREQ_64B -d $OUT/bench_WR -- python $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-verify --no-other-configs --no-host-packets > $OUT/bench_WR.log 2>&1
timeout 400 rocprofv3 --pmc TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_EA0_RDREQ_64B TCC_EA0_RDREQ_128B -d $OUT/bench_RD -- python $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-verify --no-other-configs --no-host-packets > $OUT/bench_RD.log 2>&1
timeout 400 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM -d $OUT/sq1 -- python $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-verify --no-other-configs --no-host-packets > $OUT/sq1.log 2>&1
cd $REPO
python tools/pmc_summary.py $(find $OUT/bench_WR $OUT/bench_RD $OUT/sq1 -name '*.db') --traffic $OUT/hbm_traffic.json --streams 256 --gop IPPPPPPP > $OUT/pmc.txt 2>&1; grep -v "LEVEL\|_DRAM" $OUT/pmc.txt
find $OUT -name '*.db' -size +20M -delete
fi
# capture bench (packets of a real bitstream through the reference's parser): resident and PCIe-inclusive rates
if [ -n "$CAPTURE" ]; then
python tools/make_capture.py tests/golden/streams/hd1080_ipp30.264 /tmp/hd1080_ipp30.e264 > $OUT/capture.log 2>&1
timeout 600 python bench.py --capture /tmp/hd1080_ipp30.e264 --no-cpu-baseline --steps 2 > $OUT/bench_capture.json 2> $OUT/bench_capture.err
python -c "
import json; c=json.load(open('$OUT/bench_capture.json')); print('capture', c['value'], c['bit_exact'], c['pcie_inclusive'])"
timeout 300 python bench.py --no-cpu-baseline --no-other-configs --no-verify > $OUT/bench_pcie.json 2> $OUT/bench_pcie.err
python -c "
import json; c=json.load(open('$OUT/bench_pcie.json')); print('synthetic', c['value'], c['pcie_inclusive'])"
fi
