timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_golden.py -m gpu -x -q 2>&1 | tail -2
KINDS="8 4,8,16" bash tools/gpu_ikinds.sh | grep kinds
