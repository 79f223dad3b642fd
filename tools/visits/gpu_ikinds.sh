# what one I picture of each intra kind costs (intra kernel, 256 pictures per launch); E264_I_KINDS / E264_RESIDUAL_PROB are measuring aids of bench.py
mkdir -p gpurun_out/ik
[ -n "$KINDS" ] || timeout 600 python -m pytest tests/test_hip_parity.py tests/test_hip_golden.py -m gpu -x -q 2>&1 | tail -2
B="python bench.py --no-cpu-baseline --no-other-configs --no-host-packets --no-same-input --variants 1 --gop I --steps 10 --warmup 2"
for K in ${KINDS:-4 8 16 4,8,16}; do
  E264_I_KINDS=$K timeout 300 $B > gpurun_out/ik/b_$K.json 2> gpurun_out/ik/b_$K.err
  python -c "
import json; d=json.load(open('gpurun_out/ik/b_$K.json')); print('kinds $K', d['value'], d['bit_exact'], {k.split('_')[1]:v['ms_per_launch'] for k,v in d['roofline']['kernels'].items()})"
done
