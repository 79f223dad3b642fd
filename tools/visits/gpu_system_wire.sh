#!/bin/bash
# the system figure (e264_multi: parser threads + emitters + GPU, 128 decoders on the four 1080p fixtures, no read-back) with version-4 packets and with the
# front end folding them (E264_FRONT_COMPACT=1).  usage: tools/visits/gpu_system_wire.sh TAG [threads] [loops]
TAG=${1:-syswire}; OUT=gpurun_out/$TAG; mkdir -p $OUT
T=${2:-15}; L=${3:-8}
S=tests/golden/streams
F="$S/hd1080_ipp30.264 $S/cabac_hd1080_ibbp30.264 $S/nat1080_ipp30.264 $S/cabac_nat1080_ibbp30.264"
for c in 0 1 0 1; do
  E264_FRONT_COMPACT=$c timeout 300 edge264_amd/e264_multi --front edge264_amd/libedge264_hipfront.so --hip edge264_amd/libedge264_hip.so --no-download --threads $T --repeat 32 --stay --loops $L $F 2>$OUT/err_$c.txt | tail -1 > $OUT/sys_$c.json
  python - <<PY
import json
d = json.load(open('$OUT/sys_$c.json'))
st = d.get('steady') or {}
print('compact=$c', 'whole', d['frames_per_s'], 'steady', st.get('frames_per_s'), 'decode_ms', st.get('decode_ms_per_picture'), 'avg_batch', d.get('avg_batch'))
PY
done
