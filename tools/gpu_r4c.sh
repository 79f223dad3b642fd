#!/bin/bash
# Round 4, visit C: GPU tests, the default bench line (same-input leg, staged host batches), e264_multi with device-paced batches
REPO=$(pwd); OUT=$REPO/gpurun_out/r4c; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
python -c "
import json; d=json.load(open('$OUT/bench_default.json')); print(d['value'], d['bit_exact'], {k.split('_')[1]: v['ms_per_launch'] for k,v in d['roofline']['kernels'].items()}); print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['per_core']); print('pcie', d['pcie_inclusive']['value'], d['pcie_inclusive']['pinned_in_place']['value']); print('same', json.dumps(d['same_input'])[:1500]); print('other', {k[:10]: v['value'] for k,v in d['other_configs'].items()}); print(d['per_rank']); print(d['roofline']['traffic_source'])"
M="./edge264_amd/e264_multi --front edge264_amd/libedge264_hipfront.so --hip edge264_amd/libedge264_hip.so"
S="tests/golden/streams/hd1080_ipp30.264 tests/golden/streams/cabac_hd1080_ibbp30.264"
{
for T in 15 16; do echo "== e264_multi parse-only, $T threads, 128 streams, 12 loops"; timeout 200 $M --threads $T --repeat 64 --loops 12 --parse-only $S | grep -o '"threads.*'; done
for T in 14 15 16 24; do echo "== e264_multi end to end (pinned, no read-back), $T threads, 128 streams, 12 loops"; timeout 200 $M --threads $T --repeat 64 --loops 12 --no-download $S | grep -o '"threads.*'; done
echo "== e264_multi end to end (pageable, no read-back), 15 threads"; timeout 200 $M --threads 15 --repeat 64 --loops 12 --no-download --pageable $S | grep -o '"threads.*'
echo "== e264_multi end to end WITH read-back, 15 threads"; timeout 200 $M --threads 15 --repeat 64 --loops 6 $S | grep -o '"threads.*'
echo "== 256 streams, 15 threads, no read-back"; timeout 200 $M --threads 15 --repeat 128 --loops 8 --no-download $S | grep -o '"threads.*'
} > $OUT/multi.txt 2>&1
cat $OUT/multi.txt
