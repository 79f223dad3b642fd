# intra kernel: level-size test and dependency mask hoisted -- parity, then A/B against the previous commit (I pictures and the bench GOP)
timeout 300 python -m pytest tests/test_hip_parity.py tests/test_hip_golden.py -m gpu -x -q 2>&1 | tail -2
B="python bench.py --no-cpu-baseline --no-host-packets --no-same-input --no-other-configs --no-verify --variants 1 --steps 6 --warmup 2"
for V in prev default prev default; do
  if [ $V = default ]; then unset E264_HIP_LIB; else export E264_HIP_LIB=$(pwd)/edge264_amd/variants/libedge264_hip_$V.so; fi
  timeout 100 $B --gop I > gpurun_out/r4s_$V.json 2> gpurun_out/r4s_$V.err; python -c "
import json; d=json.load(open('gpurun_out/r4s_$V.json')); print('$V gop I', d['value'], {k.split('_')[1]:v['ms_per_launch'] for k,v in d['roofline']['kernels'].items()})"
done
