#!/bin/bash
# Round 4: host-side cost per picture on the GPU box's cores (tools/hostprof) and the scaling of e264_multi's parser threads.
TAG=${1:-r4host}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
S="tests/golden/streams/hd1080_ipp30.264 tests/golden/streams/cabac_hd1080_ibbp30.264"
{
echo "== one pinned core, 5 s each: reference decoder / parser with null leaves / parser + emitters (capture sink)"
for L in oracle/_ref/libedge264_ref.so tools/hostprof/libedge264_nullfront.so edge264_amd/libedge264_hipfront.so; do ./tools/hostprof/hostprof $L 5 4 $S; done
echo "== 16 processes at once on cores 0..15 (the container's quota), parser + emitters, 5 s"
for c in $(seq 0 15); do ./tools/hostprof/hostprof edge264_amd/libedge264_hipfront.so 5 $c $S > $OUT/hp_$c.json & done; wait
cat $OUT/hp_*.json | python -c "
import sys,json; r=[json.loads(l) for l in sys.stdin]; print('16 processes: total frames/s', round(sum(x['frames_per_s'] for x in r),1), 'mean core_ms_per_picture', round(sum(x['core_ms_per_picture'] for x in r)/len(r),3), 'wall-based ms/picture/core', round(1e3*len(r)/sum(x['frames_per_s'] for x in r),3))"
echo "== 16 processes at once, reference decoder"
for c in $(seq 0 15); do ./tools/hostprof/hostprof oracle/_ref/libedge264_ref.so 5 $c $S > $OUT/hr_$c.json & done; wait
cat $OUT/hr_*.json | python -c "
import sys,json; r=[json.loads(l) for l in sys.stdin]; print('16 processes: total frames/s', round(sum(x['frames_per_s'] for x in r),1), 'mean core_ms_per_picture', round(sum(x['core_ms_per_picture'] for x in r)/len(r),3))"
M="./edge264_amd/e264_multi --front edge264_amd/libedge264_hipfront.so --hip edge264_amd/libedge264_hip.so"
for T in 1 4 8 15 16 32; do echo "== e264_multi parse-only, $T threads, 128 streams"; timeout 120 $M --threads $T --repeat 64 --loops 2 --parse-only $S | grep -o '"threads.*'; done
for T in 15 32; do echo "== e264_multi end to end (pinned packets, no read-back), $T threads, 128 streams"; timeout 120 $M --threads $T --repeat 64 --loops 4 --no-download $S | grep -o '"threads.*'; done
nproc; cat /sys/fs/cgroup/cpu.max; lscpu | grep -E "Model name|L2|L3"
} > $OUT/host.txt 2>&1
cat $OUT/host.txt
