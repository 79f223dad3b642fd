#!/bin/bash
# round 6, the visit of record: GPU tests, smoke, the default bench line (every leg), the rocprofv3 kernel trace of the timed command, the HBM-side PMC
# passes (reads / writes in separate passes, no trace domain beside them), the SQ instruction mix (text + JSON: bench.py prices the kernels against VALU
# issue with it), 512 / 1024 streams per launch.     bash tools/visits/gpu_profile_r6.sh TAG      -> gpurun_out/TAG/*: what profiles/r06_* is copied from
TAG=${1:-r06}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest_gpu.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
B="python $REPO/bench.py --no-cpu-baseline --no-verify --no-other-configs --no-host-packets --no-same-input --no-system --no-single-stream --no-staggered"
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/prof_kt -- $B > $OUT/bench_kt.json 2> $OUT/bench_kt.err; echo "kt rc=$?"
timeout 400 rocprofv3 --pmc TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_EA0_RDREQ_64B TCC_EA0_RDREQ_128B -d $OUT/pmc_rd -- $B --steps 1 --warmup 0 > /dev/null 2> $OUT/pmc_rd.err; echo "rd rc=$?"
timeout 400 rocprofv3 --pmc TCC_EA0_WRREQ TCC_EA0_WRREQ_64B -d $OUT/pmc_wr -- $B --steps 1 --warmup 0 > /dev/null 2> $OUT/pmc_wr.err; echo "wr rc=$?"
timeout 400 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM -d $OUT/pmc_sq -- $B --steps 1 --warmup 0 > /dev/null 2> $OUT/pmc_sq.err; echo "sq rc=$?"
# the host-packet legs alone (page-locked packets, version 4 and wire form): e264_expand_kernel's duration beside the four
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/prof_wire -- python $REPO/bench.py --no-cpu-baseline --no-verify --no-other-configs --no-same-input --no-system --no-single-stream --no-staggered > $OUT/bench_wire_kt.json 2> $OUT/bench_wire_kt.err; echo "wire kt rc=$?"
cd $REPO
python tools/rocprof_summary.py $(find $OUT/prof_kt -name '*.db' | head -1) > $OUT/kernel_stats.txt 2>&1; cat $OUT/kernel_stats.txt
python tools/rocprof_summary.py $(find $OUT/prof_wire -name '*.db' | head -1) > $OUT/kernel_stats_host_packets.txt 2>&1; cat $OUT/kernel_stats_host_packets.txt
python tools/pmc_summary.py $(find $OUT/pmc_rd $OUT/pmc_wr -name '*.db') --traffic $OUT/hbm_traffic.json --streams 256 --gop IPPPPPPP --bench-json $OUT/bench_kt.json > $OUT/pmc_hbm_requests.txt 2>&1
python tools/pmc_summary.py $(find $OUT/pmc_sq -name '*.db') --sq-json $OUT/sq_mix.json --streams 256 --gop IPPPPPPP --bench-json $OUT/bench_kt.json > $OUT/pmc_sq_instruction_mix.txt 2>&1
# the line of record AFTER the counters: it quotes the traffic / SQ files of this very build
cp $OUT/hbm_traffic.json profiles/r06_hbm_traffic.json; cp $OUT/sq_mix.json profiles/r06_sq_mix.json
T0=$(date +%s); timeout 1200 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$? seconds=$(( $(date +%s) - T0 ))"
for NS in 512 1024; do
  timeout 300 $B --streams $NS --steps 3 > $OUT/bench_streams$NS.json 2> $OUT/bench_streams$NS.err
  python -c "
import json; d=json.load(open('$OUT/bench_streams$NS.json')); print('streams $NS', d['value'], {k.split('_')[1]: v['ms_per_launch'] for k,v in d['roofline']['kernels'].items()})"
done
find $OUT -name '*.db' -delete
python -c "
import json; d=json.load(open('$OUT/bench_default.json')); r=d['roofline']
print(d['value'], d['bit_exact'], r['bound'], r['kernel'], r['frac'], r['valu_issue_frac'], r['end_to_end'], {k.split('_')[1]: v['ms_per_launch'] for k,v in r['kernels'].items()})
print({k.split('_')[1]: (v['valu_issue_frac'], v['lane_instr_per_sample']) for k,v in r['valu_issue']['kernels'].items()})
print(d['cpu_baseline']['value'], d['cpu_baseline'].get('per_core'))
p=d['pcie_inclusive']; print(p['value'], p.get('link_frac'), p['pinned_in_place']['value'], p['pinned_in_place'].get('link_frac'), p.get('link_h2d_GBps_measured'), p.get('pinned_wire'))
print(json.dumps(d['staggered_gops']))
print(json.dumps(d['single_stream']))
print(json.dumps(d['same_input'])[:1800]); print({k[:10]: v['value'] for k,v in d['other_configs'].items()}); print(json.dumps(d['system'])[:1200])"
python -c "
import json; t=json.load(open('$OUT/hbm_traffic.json')); print({k: (round(v['read_bytes']/1e9,3), round(v['write_bytes']/1e9,3)) for k,v in t['kernels'].items()})"
