#!/bin/bash
# end-to-end multi-stream driver: host parsing threads vs throughput (bit-exactness is covered by tests/test_multi_stream.py)
# bash tools/visits/gpu_multi.sh TAG
TAG=${1:-multi}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
nproc
timeout 600 python -m pytest tests/test_multi_stream.py -m gpu -x -q 2>&1 | tail -2
M="./edge264_amd/e264_multi --front edge264_amd/libedge264_hipfront.so --hip edge264_amd/libedge264_hip.so"
S="tests/golden/streams/hd1080_ipp30.264 tests/golden/streams/cabac_hd1080_ibbp30.264"
{
for t in 16 64 128; do echo "parse-only threads=$t"; $M --parse-only --threads $t --repeat 64 --loops 8 $S; done
for t in 16 64 128; do echo "no-download threads=$t"; $M --no-download --threads $t --repeat 64 --loops 8 $S; done
for t in 64 128; do echo "no-download threads=$t streams=256"; $M --no-download --threads $t --repeat 128 --loops 6 $S; done
for t in 64; do echo "download threads=$t"; $M --threads $t --repeat 64 --loops 4 $S; done
} 2>&1 | tee $OUT/multi.txt
