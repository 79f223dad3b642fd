timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "waves_per_frame or 1080p or max_frame" 2>&1 | tail -3
B="python bench.py --no-cpu-baseline --no-host-packets --no-same-input --no-other-configs --steps 8 --warmup 2"
for w in 8 108 107 8 108; do timeout 300 $B --waves $w > gpurun_out/r4g_$w.json 2> gpurun_out/r4g_$w.err; python -c "
import json; d=json.load(open('gpurun_out/r4g_$w.json')); print('waves $w', d['value'], d['bit_exact'], {k.split('_')[1]:v['ms_per_launch'] for k,v in d['roofline']['kernels'].items()})"; done
