#!/bin/bash
# round 5, the visit of record: GPU tests, smoke, the default bench line, the rocprofv3 kernel trace of the timed command
TAG=${1:-r5final}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest_gpu.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
B="python $REPO/bench.py --no-cpu-baseline --no-verify --no-other-configs --no-host-packets --no-same-input --no-system"
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/prof_kt -- $B > $OUT/bench_kt.json 2> $OUT/bench_kt.err; echo "kt rc=$?"
cd $REPO
python tools/rocprof_summary.py $(find $OUT/prof_kt -name '*.db' | head -1) > $OUT/kernel_stats.txt 2>&1; cat $OUT/kernel_stats.txt
find $OUT -name '*.db' -delete
python -c "
import json; d=json.load(open('$OUT/bench_default.json')); print(d['value'], d['bit_exact'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['traffic_source']['file'], {k.split('_')[1]: v['ms_per_launch'] for k,v in d['roofline']['kernels'].items()}); print(d['cpu_baseline']['value']); print(d['pcie_inclusive']['value'], d['pcie_inclusive']['pinned_in_place']['value']); print({k: v for k, v in d['same_input'].items() if k not in ('per_file', 'what')}); print({k[:10]: v['value'] for k,v in d['other_configs'].items()}); print({k: d['system'][k] for k in ('end_to_end_vs_parse_only','vs_cpu_reference_same_cores','host_cores_for_1000_streams_1080p30')}, d['system']['parse_only']['frames_per_s'], d['system']['end_to_end']['frames_per_s'])"
# the start-up of a 128-decoder run after the plain (not page-locked) mirrors of decode-to-device: loops 4 and 16 as in tools/visits/gpu_r5p.sh
M="./edge264_amd/e264_multi --front edge264_amd/libedge264_hipfront.so --hip edge264_amd/libedge264_hip.so"
S="tests/golden/streams/hd1080_ipp30.264 tests/golden/streams/cabac_hd1080_ibbp30.264 tests/golden/streams/nat1080_ipp30.264 tests/golden/streams/cabac_nat1080_ibbp30.264"
for L in 4 16; do echo "== e2e 15 stay, loops $L"; timeout 200 $M --no-download --stay --threads 15 --repeat 32 --loops $L $S 2>&1 | tail -1 | cut -c1-120,330-520; done | tee $OUT/multi_startup.txt
