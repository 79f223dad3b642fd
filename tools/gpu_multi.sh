#!/bin/bash
# end-to-end multi-stream driver: host parsing threads vs throughput (bit-exactness is covered by tests/test_multi_stream.py)
export TMPDIR=/tmp
nproc
S="tests/golden/streams/cabac_hd1080_ipp.264 tests/golden/streams/hd1080_ippb.264"
for t in 1 4 16 64; do
  ./edge264_amd/e264_multi --front oracle/_ref/libedge264_hipfront.so --hip edge264_amd/libedge264_hip.so --threads $t --repeat 32 --loops 12 $S
done
for t in 16 64; do
  ./edge264_amd/e264_multi --front oracle/_ref/libedge264_hipfront.so --hip edge264_amd/libedge264_hip.so --no-download --threads $t --repeat 32 --loops 12 $S
done
for t in 1 8 16 64; do
  ./edge264_amd/e264_multi --front oracle/_ref/libedge264_hipfront.so --hip edge264_amd/libedge264_hip.so --parse-only --threads $t --repeat 32 --loops 12 $S
done
timeout 600 python -m pytest tests/test_multi_stream.py -m gpu -x -q 2>&1 | tail -2
