#!/bin/bash
# round 5: the GPU test suite + the default bench line (every leg, the new `system` leg included)
TAG=${1:-r5f}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"; tail -3 $OUT/bench_default.err
python -c "
import json
d=json.load(open('$OUT/bench_default.json'))
print('value', d['value'], d['bit_exact'], {k[5:-7]:v['ms_per_launch'] for k,v in d['roofline']['kernels'].items()})
print('other', [(k[:10], v['value'], v['bit_exact']) for k,v in (d.get('other_configs') or {}).items()])
print('pcie', d['pcie_inclusive']['value'], d['pcie_inclusive']['pinned_in_place']['value'])
print('same', {k: v for k, v in d['same_input'].items() if k != 'what' and k != 'per_file'})
print('system', json.dumps(d['system'])[:1500])
print('cpu', {k: v for k, v in d['cpu_baseline'].items() if k != 'sample'})
"
