S=tests/golden/streams
F="$S/hd1080_ipp30.264 $S/cabac_hd1080_ibbp30.264 $S/nat1080_ipp30.264 $S/cabac_nat1080_ibbp30.264"
for rep in 1 2; do for t in 4 8 15; do
  E264_GATHER_THREADS=$t timeout 300 edge264_amd/e264_multi --front edge264_amd/libedge264_hipfront.so --hip edge264_amd/libedge264_hip.so --no-download --threads 15 --repeat 32 --stay --loops 8 $F 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.load(sys.stdin); st=d.get('steady') or {}; print('gather', $t, 'whole', d['frames_per_s'], 'steady', st.get('frames_per_s'), 'decode_ms', st.get('decode_ms_per_picture'))"
done; done
for t in 4 8 15; do E264_GATHER_THREADS=$t timeout 300 python tools/pin_probe.py nat1080_ipp30.264 pinned_trusted_own_buffers,pinned_trusted,pinned_trusted_own_buffers | python3 -c "
import json,sys; d=json.load(sys.stdin); print('probe gather', $t, {k:[x['frames_per_s'] for x in v] for k,v in d.items() if isinstance(v,list)})"; done
