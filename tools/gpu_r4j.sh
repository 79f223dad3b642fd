# intra kernel: extra distance between the rows of the wavefront (E264_INTRA_SLACK)
B="python bench.py --no-cpu-baseline --no-other-configs --no-host-packets --no-same-input --variants 1 --steps 10 --warmup 2"
for S in 0 2 4 8 0; do
  if [ $S = 0 ]; then unset E264_HIP_LIB; else export E264_HIP_LIB=$(pwd)/edge264_amd/variants/libedge264_hip_slack$S.so; fi
  for G in I IPPPPPPP; do timeout 300 $B --gop $G > gpurun_out/r4j_$S$G.json 2> gpurun_out/r4j_$S$G.err; python -c "
import json; d=json.load(open('gpurun_out/r4j_$S$G.json')); print('slack $S gop $G', d['value'], d['bit_exact'], {k.split('_')[1]:v['ms_per_launch'] for k,v in d['roofline']['kernels'].items()})"; done; done
