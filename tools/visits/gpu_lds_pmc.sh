# LDS pipeline of the kernels: active cycles and bank-conflict cycles (SQ counters), bench GOP and I pictures
OUT=$(pwd)/gpurun_out/ldspmc; mkdir -p $OUT; export TMPDIR=/tmp
R=$(pwd)
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_LDS[A-Z_]*\|SQ_ACTIVE_INST_LDS\|SQ_INST_CYCLES_[A-Z_]*\|SQ_INSTS_LDS" | sort -u | tr '\n' ' '; echo
B="python $R/bench.py --no-cpu-baseline --no-verify --no-other-configs --no-host-packets --no-same-input --variants 1 --steps 1 --warmup 0"
cd /tmp
for G in I IPPPPPPP; do
  timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL -d $OUT/g$G -- $B --gop $G > /dev/null 2> $OUT/g$G.err
  echo "== gop $G"; python $R/tools/pmc_summary.py $(find $OUT/g$G -name '*.db') 2>&1 | grep -v "^\s*$"
done
cd $R; find $OUT -name '*.db' -size +5M -delete
