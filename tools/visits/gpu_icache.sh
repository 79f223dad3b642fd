# instruction-cache counters of the four kernels (is the 139-KB prediction kernel fetch-bound?)
OUT=$(pwd)/gpurun_out/icache; mkdir -p $OUT; export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_WAIT_INST[A-Z_]*\|SQC_INST[A-Z_]*\|SQ_INST_CYCLES[A-Z_]*\|SQ_ACTIVE_INST[A-Z_]*\|SQ_INST_LEVEL[A-Z_]*" | sort -u > $OUT/avail.txt; cat $OUT/avail.txt | tr '\n' ' '; echo
B="python $(pwd)/bench.py --no-cpu-baseline --no-verify --no-other-configs --no-host-packets --no-same-input --variants 1 --steps 1 --warmup 0"
R=$(pwd)
cd /tmp
timeout 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE -d $OUT/p1 -- $B > /dev/null 2> $OUT/p1.err; echo "p1 rc=$?"
timeout 300 rocprofv3 --pmc SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU -d $OUT/p2 -- $B > /dev/null 2> $OUT/p2.err; echo "p2 rc=$?"
cd $R
python tools/pmc_summary.py $(find $OUT/p1 $OUT/p2 -name '*.db') > $OUT/summary.txt 2>&1; cat $OUT/summary.txt
find $OUT -name '*.db' -size +20M -delete
