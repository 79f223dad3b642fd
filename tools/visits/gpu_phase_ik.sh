# phase profile of the intra kernel: I pictures of one intra kind each, and the bench GOP (I + 7 P: the P pictures' share = GOP - I at the same number of steps)
export E264_PHASE_INTRA=1 E264_HIP_LIB=$(pwd)/edge264_amd/variants/libedge264_hip_phasei.so
for K in ${KINDS:-16 8 4}; do echo "== kinds $K, gop I"; E264_I_KINDS=$K bash tools/visits/gpu_phase2.sh phik_$K --gop I --variants 1 --no-other-configs --no-host-packets --no-same-input | grep -A12 "^e264_deblock_kernel" | tail -12; done
echo "== all kinds, gop I"; bash tools/visits/gpu_phase2.sh phik_I --gop I --variants 1 --no-other-configs --no-host-packets --no-same-input | grep -A12 "^e264_deblock_kernel" | tail -12
echo "== all kinds, gop IPPPPPPP"; bash tools/visits/gpu_phase2.sh phik_IP --gop IPPPPPPP --variants 1 --no-other-configs --no-host-packets --no-same-input | grep -A12 "^e264_deblock_kernel" | tail -12
