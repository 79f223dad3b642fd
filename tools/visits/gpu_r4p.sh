timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_golden.py -m gpu -x -q 2>&1 | tail -2
for V in ihead default ihead default; do
  if [ $V = default ]; then unset E264_HIP_LIB; else export E264_HIP_LIB=$(pwd)/edge264_amd/variants/libedge264_hip_$V.so; fi
  echo "== $V"; KINDS="16 4 8" bash tools/visits/gpu_ikinds.sh | grep kinds
done
