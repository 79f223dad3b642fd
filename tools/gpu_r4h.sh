# default bench with the GOP variants (4 distinct GOPs dealt to the streams), timed by the wall clock too
mkdir -p gpurun_out/r4h
S=$(date +%s); timeout 900 python bench.py > gpurun_out/r4h/bench_default.json 2> gpurun_out/r4h/bench_default.err; echo "bench rc=$? wall=$(( $(date +%s) - S ))s"
python - <<'PY'
import json
d = json.load(open('gpurun_out/r4h/bench_default.json'))
print(d['value'], d['bit_exact'], d['verify'], {k.split('_')[1]: v['ms_per_launch'] for k, v in d['roofline']['kernels'].items()})
print('pcie', d['pcie_inclusive']['value'], 'cpu', d['cpu_baseline']['value'], 'same', d['same_input']['gpu_resident_frames_per_s'], d['same_input']['bit_exact'])
print(d['config']['workload'][:400])
PY
S=$(date +%s); timeout 600 python bench.py --variants 1 --no-cpu-baseline --no-host-packets --no-same-input --no-other-configs > gpurun_out/r4h/bench_v1.json 2> gpurun_out/r4h/bench_v1.err; echo "v1 rc=$? wall=$(( $(date +%s) - S ))s"
python -c "
import json; d=json.load(open('gpurun_out/r4h/bench_v1.json')); print('variants 1', d['value'], d['bit_exact'], {k.split('_')[1]:v['ms_per_launch'] for k,v in d['roofline']['kernels'].items()})"
