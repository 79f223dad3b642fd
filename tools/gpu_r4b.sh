timeout 600 python -m pytest tests/test_hip_backend.py tests/test_multi_stream.py tests/test_replay.py -m gpu -x -q 2>&1 | tail -3
timeout 300 python bench.py --no-cpu-baseline --no-other-configs --steps 6 --warmup 2 > gpurun_out/r4b_bench.json 2> gpurun_out/r4b_bench.err; python -c "
import json; d=json.load(open('gpurun_out/r4b_bench.json')); print(d['value'], d['bit_exact'], d['pcie_inclusive'])"
bash tools/gpu_r4_host.sh r4host
