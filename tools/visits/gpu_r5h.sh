#!/bin/bash
# round 5: packets assembled in the thread's scratch + streaming copy (default) against assembled in place in page-locked memory (E264_FRONT_IN_PLACE=1)
TAG=${1:-r5h}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_multi_stream.py -m gpu -x -q 2>&1 | tail -2
M="./edge264_amd/e264_multi --front edge264_amd/libedge264_hipfront.so --hip edge264_amd/libedge264_hip.so"
S="tests/golden/streams/hd1080_ipp30.264 tests/golden/streams/cabac_hd1080_ibbp30.264"
run() { echo "== $1"; shift; timeout 120 "$@" 2>&1 | tail -1 | python -c "
import sys, json
l = sys.stdin.read().strip()
try:
    d = json.loads(l); print({k: d[k] for k in ('threads', 'frames_per_s', 'decode_ms_per_picture', 'avg_batch', 'thread_seconds')})
except Exception as e:
    print('FAILED', l[-300:])"; }
{
run "parse-only 15" $M --parse-only --threads 15 --repeat 64 --loops 10 $S
for rep in 1 2; do
E264_FRONT_IN_PLACE=1 run "e2e 15 in place" $M --no-download --threads 15 --repeat 64 --loops 8 $S
run "e2e 15 scratch + streaming" $M --no-download --threads 15 --repeat 64 --loops 8 $S
done
E264_FRONT_IN_PLACE=1 run "e2e 15 in place, 256 decoders" $M --no-download --threads 15 --repeat 128 --loops 4 $S
run "e2e 15 scratch + streaming, 256 decoders" $M --no-download --threads 15 --repeat 128 --loops 4 $S
run "e2e 15 scratch + streaming, 64 decoders" $M --no-download --threads 15 --repeat 32 --loops 16 $S
run "parse-only 15, 64 decoders" $M --parse-only --threads 15 --repeat 32 --loops 20 $S
run "parse-only 15, 256 decoders" $M --parse-only --threads 15 --repeat 128 --loops 5 $S
} 2>&1 | tee $OUT/multi.txt
