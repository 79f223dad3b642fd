#!/bin/bash
# round 5: is the end-to-end deficit a warm-up cost (device frames, page-locked mirrors and packet buffers of 128 decoders allocated inside the clock)?
TAG=${1:-r5p}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
M="./edge264_amd/e264_multi --front edge264_amd/libedge264_hipfront.so --hip edge264_amd/libedge264_hip.so"
S="tests/golden/streams/hd1080_ipp30.264 tests/golden/streams/cabac_hd1080_ibbp30.264 tests/golden/streams/nat1080_ipp30.264 tests/golden/streams/cabac_nat1080_ibbp30.264"
run() { echo "== $1"; shift; timeout 200 "$@" 2>&1 | tail -1 | python -c "
import sys, json
l = sys.stdin.read().strip()
try:
    d = json.loads(l); print({k: d[k] for k in ('threads', 'frames', 'seconds', 'frames_per_s', 'decode_ms_per_picture', 'avg_batch')})
except Exception as e:
    print('FAILED', l[-300:])"; }
{
for L in 4 16 48; do run "e2e 15 stay, loops $L" $M --no-download --stay --threads 15 --repeat 32 --loops $L $S; done
for L in 4 16 48; do run "parse-only 15 stay, loops $L" $M --parse-only --stay --threads 15 --repeat 32 --loops $L $S; done
} 2>&1 | tee $OUT/multi.txt
