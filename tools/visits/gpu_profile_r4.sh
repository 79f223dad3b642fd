#!/bin/bash
# One GPU-box visit that produces what profiles/r04_* is made of: the GPU tests, the default bench line, the rocprofv3 kernel
# trace of the same command, the HBM-side PMC passes (reads / writes in separate passes, no trace domain beside them), the SQ
# instruction mix, stream-count variants, e264_multi end to end, the host cost per picture.   bash tools/visits/gpu_profile_r4.sh TAG
TAG=${1:-r04}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest_gpu.log
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
B="python $REPO/bench.py --no-cpu-baseline --no-verify --no-other-configs --no-host-packets --no-same-input"
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/prof_kt -- $B > $OUT/bench_kt.json 2> $OUT/bench_kt.err; echo "kt rc=$?"
timeout 400 rocprofv3 --pmc TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_EA0_RDREQ_64B TCC_EA0_RDREQ_128B -d $OUT/pmc_rd -- $B --steps 1 --warmup 0 > /dev/null 2> $OUT/pmc_rd.err; echo "rd rc=$?"
timeout 400 rocprofv3 --pmc TCC_EA0_WRREQ TCC_EA0_WRREQ_64B -d $OUT/pmc_wr -- $B --steps 1 --warmup 0 > /dev/null 2> $OUT/pmc_wr.err; echo "wr rc=$?"
timeout 400 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM -d $OUT/pmc_sq -- $B --steps 1 --warmup 0 > /dev/null 2> $OUT/pmc_sq.err; echo "sq rc=$?"
cd $REPO
python tools/rocprof_summary.py $(find $OUT/prof_kt -name '*.db' | head -1) > $OUT/kernel_stats.txt 2>&1; cat $OUT/kernel_stats.txt
python tools/pmc_summary.py $(find $OUT/pmc_rd $OUT/pmc_wr -name '*.db') --traffic $OUT/hbm_traffic.json --streams 256 --gop IPPPPPPP --bench-json $OUT/bench_default.json > $OUT/pmc_hbm_requests.txt 2>&1
python tools/pmc_summary.py $(find $OUT/pmc_sq -name '*.db') > $OUT/pmc_sq_instruction_mix.txt 2>&1
for NS in 512 1024; do
  timeout 300 $B --streams $NS --steps 3 > $OUT/bench_streams$NS.json 2> $OUT/bench_streams$NS.err
  python -c "
import json; d=json.load(open('$OUT/bench_streams$NS.json')); print('streams $NS', d['value'], {k.split('_')[1]: v['ms_per_launch'] for k,v in d['roofline']['kernels'].items()})"
done
timeout 300 $B --waves 108 > $OUT/bench_split_waves.json 2> $OUT/bench_split_waves.err
python -c "
import json; d=json.load(open('$OUT/bench_split_waves.json')); print('deblock split kernel (8 luma / chroma waves)', d['value'], {k.split('_')[1]: v['ms_per_launch'] for k,v in d['roofline']['kernels'].items()})"
find $OUT -name '*.db' -size +20M -delete
python -c "
import json; d=json.load(open('$OUT/bench_default.json')); print(d['value'], d['bit_exact'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['end_to_end'], {k.split('_')[1]: v['ms_per_launch'] for k,v in d['roofline']['kernels'].items()}); print(d['cpu_baseline']['value'], d['cpu_baseline']['per_core'], d['cpu_baseline']['single_process']); print(d['pcie_inclusive']['value'], d['pcie_inclusive']['pinned_in_place']['value']); print(json.dumps(d['same_input'])[:900]); print({k[:10]: v['value'] for k,v in d['other_configs'].items()})"
python -c "
import json; t=json.load(open('$OUT/hbm_traffic.json')); print({k: (round(v['read_bytes']/1e9,3), round(v['write_bytes']/1e9,3)) for k,v in t['kernels'].items()})"
# the multi-stream driver end to end (reference parser + emitters + GPU) and the host cost per picture
M="./edge264_amd/e264_multi --front edge264_amd/libedge264_hipfront.so --hip edge264_amd/libedge264_hip.so"
S="tests/golden/streams/hd1080_ipp30.264 tests/golden/streams/cabac_hd1080_ibbp30.264"
{
echo "== one pinned core, 5 s each: reference decoder / parser with null leaves / parser + emitters (capture sink)"
for L in oracle/_ref/libedge264_ref.so tools/hostprof/libedge264_nullfront.so edge264_amd/libedge264_hipfront.so; do ./tools/hostprof/hostprof $L 5 4 $S; done
echo "== 16 processes at once on cores 0..15 (the container's quota), parser + emitters, 5 s"
for c in $(seq 0 15); do ./tools/hostprof/hostprof edge264_amd/libedge264_hipfront.so 5 $c $S > $OUT/hp_$c.json & done; wait
cat $OUT/hp_*.json | python -c "
import sys,json; r=[json.loads(l) for l in sys.stdin]; print('16 processes: total frames/s', round(sum(x['frames_per_s'] for x in r),1), 'mean core_ms_per_picture', round(sum(x['core_ms_per_picture'] for x in r)/len(r),3))"
echo "== 16 processes at once, reference decoder"
for c in $(seq 0 15); do ./tools/hostprof/hostprof oracle/_ref/libedge264_ref.so 5 $c $S > $OUT/hr_$c.json & done; wait
cat $OUT/hr_*.json | python -c "
import sys,json; r=[json.loads(l) for l in sys.stdin]; print('16 processes: total frames/s', round(sum(x['frames_per_s'] for x in r),1), 'mean core_ms_per_picture', round(sum(x['core_ms_per_picture'] for x in r)/len(r),3))"
for T in 15; do echo "== e264_multi parse-only, $T threads, 128 streams, 12 loops"; timeout 200 $M --threads $T --repeat 64 --loops 12 --parse-only --pin $S | grep -o '"threads.*'; done
for T in 13 15; do echo "== e264_multi end to end (page-locked packets in place, no read-back), $T threads, 128 streams, 12 loops"; timeout 200 $M --threads $T --repeat 64 --loops 12 --no-download $S | grep -o '"threads.*'; done
echo "== e264_multi end to end WITH read-back, 15 threads"; timeout 200 $M --threads 15 --repeat 64 --loops 6 $S | grep -o '"threads.*'
echo "== emitter profile (rdtsc counters, instrumented front end: each call costs ~25 cycles more), e2e, 8 threads"; timeout 200 ./edge264_amd/e264_multi --front tools/hostprof/libedge264_front_prof.so --hip edge264_amd/libedge264_hip.so --threads 8 --repeat 64 --loops 4 --no-download $S 2> $OUT/prof.err | grep -o '"threads.*'
python -c "
import sys,re,collections
c=collections.Counter(); n=collections.Counter()
for l in open('$OUT/prof.err'):
    m=re.match(r'emit-profile (.{20})\s+calls\s+(\d+)\s+cycles\s+(\d+)',l)
    if m: c[m.group(1).strip()]+=int(m.group(3)); n[m.group(1).strip()]+=int(m.group(2))
for k in c: print(f'   {k:22s} calls {n[k]:10d}  Mcycles {c[k]/1e6:10.1f}  per call {c[k]/max(n[k],1):8.0f}')"
nproc; cat /sys/fs/cgroup/cpu.max; lscpu | grep -E "Model name|L2|L3"
} > $OUT/host.txt 2>&1
cat $OUT/host.txt
