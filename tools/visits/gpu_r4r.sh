# chroma sink tests hoisted: parity, A/B against the previous commit, then the default bench line + kernel trace of the final tree
OUT=$(pwd)/gpurun_out/r04y; mkdir -p $OUT; export TMPDIR=/tmp; REPO=$(pwd)
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_golden.py tests/test_hip_backend.py -m gpu -x -q 2>&1 | tail -2
B="python bench.py --no-cpu-baseline --no-host-packets --no-same-input --no-other-configs --variants 1 --steps 8 --warmup 2"
for V in prev default prev default; do
  if [ $V = default ]; then unset E264_HIP_LIB; else export E264_HIP_LIB=$(pwd)/edge264_amd/variants/libedge264_hip_$V.so; fi
  timeout 300 $B > gpurun_out/r4r_$V.json 2> gpurun_out/r4r_$V.err; python -c "
import json; d=json.load(open('gpurun_out/r4r_$V.json')); print('$V', d['value'], d['bit_exact'], {k.split('_')[1]:v['ms_per_launch'] for k,v in d['roofline']['kernels'].items()})"
done
unset E264_HIP_LIB
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/prof_kt -- python $REPO/bench.py --no-cpu-baseline --no-verify --no-other-configs --no-host-packets --no-same-input > $OUT/bench_kt.json 2> $OUT/bench_kt.err; echo "kt rc=$?"
cd $REPO
python tools/rocprof_summary.py $(find $OUT/prof_kt -name '*.db' | head -1) > $OUT/kernel_stats.txt 2>&1; head -6 $OUT/kernel_stats.txt
find $OUT -name '*.db' -size +20M -delete
python -c "
import json; d=json.load(open('$OUT/bench_default.json')); print(d['value'], d['bit_exact'], d['roofline']['frac'], d['roofline']['end_to_end'], {k.split('_')[1]: v['ms_per_launch'] for k,v in d['roofline']['kernels'].items()}); print(d['cpu_baseline']['value'], d['pcie_inclusive']['value'], d['pcie_inclusive']['pinned_in_place']['value'], d['same_input']['gpu_resident_frames_per_s'], d['same_input']['gpu_pcie_inclusive_frames_per_s'], {k[:10]: v['value'] for k,v in d['other_configs'].items()}, d['roofline']['traffic_source']['stale'])"
