#!/bin/bash
# e264_multi end to end on the GPU box (reference parser + emitters + back end), 128 streams of the two 1080p fixtures:
# packets assembled in page-locked buffers and submitted in place (default) against pageable packets validated + copied
# by the back end (--pageable), and the parser alone.   bash tools/visits/gpu_multi_r3.sh TAG
TAG=${1:-multi}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest_gpu.log
M="./edge264_amd/e264_multi --front edge264_amd/libedge264_hipfront.so --hip edge264_amd/libedge264_hip.so"
S="tests/golden/streams/hd1080_ipp30.264 tests/golden/streams/cabac_hd1080_ibbp30.264"
: > $OUT/multi.txt
for T in 16 32 64; do
  for MODE in "" "--pageable"; do
    { echo "e264_multi 128 streams, $T threads, no read-back ${MODE:---pinned (default)}"; timeout 300 $M --threads $T --repeat 64 --loops 8 --no-download $MODE $S; } >> $OUT/multi.txt 2>&1
  done
done
{ echo "e264_multi 128 streams, 32 threads, parse only (no device)"; timeout 300 $M --threads 32 --repeat 64 --loops 8 --parse-only $S; } >> $OUT/multi.txt 2>&1
{ echo "e264_multi 128 streams, 32 threads, WITH read-back of every frame"; timeout 300 $M --threads 32 --repeat 64 --loops 4 $S; } >> $OUT/multi.txt 2>&1
cat $OUT/multi.txt
