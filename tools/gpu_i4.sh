#!/bin/bash
# intra kernel variants on the mixed-kind I picture, the 4x4-only I picture (configs[1]) and the default GOP
TAG=${1:-i4}; REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
run() { # label lib
  for gop in IIII IPPPPPPP; do
    E264_HIP_LIB=$2 timeout 300 python bench.py --no-cpu-baseline --no-host-packets --gop $gop --steps 3 > $OUT/b.json 2>/dev/null
    python -c "
import json,sys; d=json.load(open(sys.argv[1])); o=d.get('other_configs') or {}
print(sys.argv[2], sys.argv[3], d['value'], d['bit_exact'], 'intra', d['roofline']['kernels']['e264_intra_kernel']['ms_per_launch'], {k[:10]: v['kernel_ms_per_launch']['e264_intra_kernel'] for k,v in o.items()})" $OUT/b.json $1 $gop
  done
}
run main $REPO/edge264_amd/libedge264_hip.so
for lib in edge264_amd/variants/*.so; do run $(basename $lib .so) $REPO/$lib; done
