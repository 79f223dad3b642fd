#!/bin/bash
# round 5: where does end to end lose against parse-only?  e264_multi, 128 decoders of the two 1080p fixtures, one setting changed at a time
TAG=${1:-r5g}; OUT=gpurun_out/$TAG; mkdir -p $OUT
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
nproc; cat /sys/fs/cgroup/cpu.max
run "parse-only 15" $M --parse-only --threads 15 --repeat 64 --loops 10 $S
run "parse-only 16" $M --parse-only --threads 16 --repeat 64 --loops 10 $S
for t in 12 13 14 15 16; do run "e2e threads=$t" $M --no-download --threads $t --repeat 64 --loops 8 $S; done
run "e2e 15 pin" $M --no-download --pin --threads 15 --repeat 64 --loops 8 $S
run "e2e 14 pin" $M --no-download --pin --threads 14 --repeat 64 --loops 8 $S
for a in 1 2 6; do run "e2e 15 ahead=$a" $M --no-download --ahead $a --threads 15 --repeat 64 --loops 8 $S; done
run "e2e 15 pageable" $M --no-download --pageable --threads 15 --repeat 64 --loops 8 $S
E264_HOST_THREADS=0 run "e2e 15 pageable, no host pool" $M --no-download --pageable --threads 15 --repeat 64 --loops 8 $S
run "e2e 15, 64 decoders" $M --no-download --threads 15 --repeat 32 --loops 16 $S
run "e2e 15, 256 decoders" $M --no-download --threads 15 --repeat 128 --loops 4 $S
} 2>&1 | tee $OUT/multi.txt
