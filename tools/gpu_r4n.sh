timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_golden.py -m gpu -x -q 2>&1 | tail -2
KINDS="16 4 8 4,8,16" bash tools/gpu_ikinds.sh | grep kinds
B="python bench.py --no-cpu-baseline --no-host-packets --no-same-input --steps 8 --warmup 2"
timeout 300 $B > gpurun_out/r4n.json 2> gpurun_out/r4n.err; python -c "
import json; d=json.load(open('gpurun_out/r4n.json')); print('headline', d['value'], d['bit_exact'], {k.split('_')[1]:v['ms_per_launch'] for k,v in d['roofline']['kernels'].items()}, {k[:10]: v['value'] for k,v in d.get('other_configs',{}).items()})"
