#!/bin/bash
# wire packets (version 5) on the device: their -m gpu tests, then the bench's same-input leg, which times the front end's road with version-4 and with
# folded packets on the same four 1080p files (gpu_pcie_inclusive_pinned_frames_per_s / ..._wire_frames_per_s).  usage: tools/visits/gpu_wire.sh TAG [all]
TAG=${1:-wire}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_hip_wire.py -x -q > $OUT/pytest_wire.log 2>&1; echo "pytest wire rc=$?"; tail -4 $OUT/pytest_wire.log
if [ -n "$2" ]; then
  timeout 1200 python -m pytest tests -m gpu -x -q --deselect tests/test_hip_wire.py > $OUT/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?"; tail -3 $OUT/pytest_gpu.log
fi
timeout 900 python bench.py --no-cpu-baseline --no-other-configs --no-system --no-single-stream --no-staggered > $OUT/bench.json 2> $OUT/bench.err
python - <<PY
import json
try:
    d = json.load(open('$OUT/bench.json'))
    print('headline', d['value'], d['bit_exact'])
    pc = d['pcie_inclusive']
    print('pcie_inclusive', pc['value'], pc.get('link_frac'), 'pinned', pc['pinned_in_place']['value'], pc['pinned_in_place'].get('link_frac'), 'wire', pc['pinned_wire'])
    s = d['same_input']
    print('same_input bit_exact', s['bit_exact'], {k: v for k, v in s.items() if k.startswith('gpu_') or k.startswith('wire')})
    for f, r in s['per_file'].items():
        print('  ', f, {k: r[k] for k in ('packet_MB_per_picture', 'wire_MB_per_picture', 'gpu_resident_frames_per_s', 'gpu_pcie_inclusive_pinned_frames_per_s', 'gpu_pcie_inclusive_pinned_wire_frames_per_s', 'host_parse_emit_frames_per_s_one_core', 'host_parse_emit_wire_frames_per_s_one_core', 'mismatching')})
except Exception as e:
    print('FAILED', e); print(open('$OUT/bench.err').read()[-2500:])
PY
