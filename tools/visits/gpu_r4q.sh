# pred kernel: sink_row tests merged, 8x8 dequant branch-free, plain multiplies -- A/B against the previous commit on one box
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_golden.py -m gpu -x -q 2>&1 | tail -2
B="python bench.py --no-cpu-baseline --no-host-packets --no-same-input --variants 1 --steps 8 --warmup 2"
for V in prev default prev default; do
  if [ $V = default ]; then unset E264_HIP_LIB; else export E264_HIP_LIB=$(pwd)/edge264_amd/variants/libedge264_hip_$V.so; fi
  timeout 300 $B > gpurun_out/r4q_$V.json 2> gpurun_out/r4q_$V.err; python -c "
import json; d=json.load(open('gpurun_out/r4q_$V.json')); print('$V', d['value'], d['bit_exact'], {k.split('_')[1]:v['ms_per_launch'] for k,v in d['roofline']['kernels'].items()}, {k[:10]: v['value'] for k,v in d.get('other_configs',{}).items()})"
done
