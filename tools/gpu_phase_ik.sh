# phase profile of the intra kernel on I pictures of one intra kind each
export E264_PHASE_INTRA=1 E264_HIP_LIB=$(pwd)/edge264_amd/variants/libedge264_hip_phasei.so
for K in 16 8 4; do echo "== kinds $K"; E264_I_KINDS=$K bash tools/gpu_phase2.sh phik_$K --gop I --variants 1 --no-other-configs --no-host-packets --no-same-input | grep -A12 "^e264_deblock_kernel" | tail -11; done
