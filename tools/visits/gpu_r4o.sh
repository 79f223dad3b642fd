# A/B of the residual changes: HEAD (3.26 / 2.73 / 2.32) / new idct4x4 flow + old level_at / both new
for V in ihead lvlbr default ihead; do
  if [ $V = default ]; then unset E264_HIP_LIB; else export E264_HIP_LIB=$(pwd)/edge264_amd/variants/libedge264_hip_$V.so; fi
  echo "== $V"; KINDS="16 4 8" bash tools/visits/gpu_ikinds.sh | grep kinds
done
