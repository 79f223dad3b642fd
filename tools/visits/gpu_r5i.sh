#!/bin/bash
# round 5: e264_multi --stay (a thread parses up to `ahead` pictures of one decoder in a row) against the rotation after every picture
TAG=${1:-r5i}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
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
run "parse-only 15 stay ahead 8" $M --parse-only --stay --ahead 8 --threads 15 --repeat 64 --loops 10 $S
run "e2e 15" $M --no-download --threads 15 --repeat 64 --loops 8 $S
run "e2e 15 stay (ahead 3)" $M --no-download --stay --threads 15 --repeat 64 --loops 8 $S
run "e2e 15 stay ahead 6" $M --no-download --stay --ahead 6 --threads 15 --repeat 64 --loops 8 $S
run "e2e 15 stay ahead 12" $M --no-download --stay --ahead 12 --threads 15 --repeat 64 --loops 8 $S
run "e2e 15" $M --no-download --threads 15 --repeat 64 --loops 8 $S
run "e2e 15 stay ahead 6, 256 decoders" $M --no-download --stay --ahead 6 --threads 15 --repeat 128 --loops 4 $S
run "e2e 15 stay ahead 30 (a whole stream per visit)" $M --no-download --stay --ahead 30 --threads 15 --repeat 64 --loops 8 $S
} 2>&1 | tee $OUT/multi.txt
