# intra kernel timing ablations: no wait for the row above / no release fence before the progress store (both: wrong samples)
export E264_ALLOW_ABLATION=1
B="python bench.py --no-cpu-baseline --no-other-configs --no-host-packets --no-same-input --no-verify --variants 1 --steps 10 --warmup 2"
for V in default inowait inofence default; do
  if [ $V = default ]; then unset E264_HIP_LIB; else export E264_HIP_LIB=$(pwd)/edge264_amd/variants/libedge264_hip_$V.so; fi
  for G in I IPPPPPPP; do for W in 16 8; do timeout 300 $B --gop $G --intra-waves $W > gpurun_out/r4k_$V$G$W.json 2> gpurun_out/r4k_$V$G$W.err; python -c "
import json; d=json.load(open('gpurun_out/r4k_$V$G$W.json')); print('$V gop $G intra waves $W', d['value'], {k.split('_')[1]:v['ms_per_launch'] for k,v in d['roofline']['kernels'].items()})"; done; done; done
