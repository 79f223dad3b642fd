#!/bin/bash
# round 5: per-kernel times on the bitstream fixtures (same_input leg), default library against variants
# usage: tools/visits/gpu_r5j.sh TAG "main nozero ..."
TAG=${1:-r5j}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
REPO=$(pwd)
for v in $2; do
  lib=$REPO/edge264_amd/variants/libedge264_hip_$v.so
  [ $v = main ] && lib=$REPO/edge264_amd/libedge264_hip.so
  E264_ALLOW_ABLATION=1 E264_HIP_LIB=$lib timeout 400 python bench.py --no-cpu-baseline --no-host-packets --no-other-configs --no-system $BENCH_ARGS > $OUT/bench_$v.json 2> $OUT/bench_$v.err
  python -c "
import json,sys
d=json.load(open('$OUT/bench_$v.json'))
print('$v', 'headline', d['value'], d['bit_exact'], {k[5:-7]:v['ms_per_launch'] for k,v in d['roofline']['kernels'].items()})
s=d['same_input']
print('  same_input bit_exact', s.get('bit_exact'), 'resident', s.get('gpu_resident_frames_per_s'), 'pcie', s.get('gpu_pcie_inclusive_frames_per_s'))
for f, r in s['per_file'].items():
    print('   ', f, r['gpu_resident_frames_per_s'], {k[5:-7]: v for k, v in r['kernel_ms_per_launch'].items()}, 'mismatching', r['mismatching'])
"
done
