#!/bin/bash
# where does the host side of e264_multi stop scaling: one process with T threads vs P processes with T/P threads (parse only, no GPU)
TAG=${1:-multi}; OUT=gpurun_out/$TAG; mkdir -p $OUT
M="./edge264_amd/e264_multi --front edge264_amd/libedge264_hipfront.so --hip edge264_amd/libedge264_hip.so --parse-only"
S="tests/golden/streams/hd1080_ipp30.264 tests/golden/streams/cabac_hd1080_ibbp30.264"
{
echo "1 process x 64 threads"; $M --threads 64 --repeat 64 --loops 8 $S
echo "4 processes x 16 threads (32 streams each)"
for p in 1 2 3 4; do $M --threads 16 --repeat 16 --loops 8 $S & done; wait
echo "8 processes x 16 threads"
for p in 1 2 3 4 5 6 7 8; do $M --threads 16 --repeat 16 --loops 8 $S & done; wait
echo "1 process x 64 threads, ahead 8"; $M --threads 64 --repeat 64 --loops 8 --ahead 8 $S
which perf strace ltrace 2>&1 | head -3
} 2>&1 | tee $OUT/multi2.txt
