#!/bin/bash
# round 6 A/B runner: (optional) GPU tests, then for the default library and each named variant the headline bench with per-kernel times and the
# same-input leg (the four 1080p bitstream fixtures, per-kernel times per file).  Every frame is verified (bit_exact / mismatching in the output).
# usage: tools/visits/gpu_r6.sh TAG "main variantA variantB ..." [tests]      (variant X = edge264_amd/variants/libedge264_hip_X.so)
#        a spec may carry environment settings: "main@E264_SIDE_QUEUE=2"
#        BENCH_ARGS adds bench arguments (e.g. --no-same-input)
TAG=${1:-r6}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
REPO=$(pwd)
if [ -n "$3" ]; then
  timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
fi
for spec in $2; do   # spec = variant[@ENV=VAL[,ENV=VAL...]]
  v=${spec%%@*}; envs=""; [ "$spec" != "$v" ] && envs=$(echo "${spec#*@}" | tr ',' ' ')
  lib=$REPO/edge264_amd/variants/libedge264_hip_$v.so
  [ $v = main ] && lib=$REPO/edge264_amd/libedge264_hip.so
  v=$(echo "$spec" | tr '@=,' '___')
  env $envs E264_ALLOW_ABLATION=1 E264_HIP_LIB=$lib timeout 500 python bench.py --no-cpu-baseline --no-host-packets --no-other-configs --no-system --no-staggered $BENCH_ARGS > $OUT/bench_$v.json 2> $OUT/bench_$v.err
  python - <<PY
import json
try:
    d = json.load(open('$OUT/bench_$v.json'))
    print('$v', 'headline', d['value'], d['bit_exact'], {k[5:-7]: v['ms_per_launch'] for k, v in d['roofline']['kernels'].items()})
    s = d.get('same_input')
    if s:
        print('  same_input bit_exact', s.get('bit_exact'), 'resident', s.get('gpu_resident_frames_per_s'), 'pcie', s.get('gpu_pcie_inclusive_frames_per_s'))
        for f, r in s['per_file'].items():
            print('   ', f, r['gpu_resident_frames_per_s'], {k[5:-7]: v for k, v in r['kernel_ms_per_launch'].items()}, 'mismatching', r['mismatching'])
except Exception as e:
    print('$v', 'FAILED', e); print(open('$OUT/bench_$v.err').read()[-1500:])
PY
done
