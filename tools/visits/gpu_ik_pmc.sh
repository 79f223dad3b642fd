# instructions per macroblock of the intra kernel, by kind (SQ counters on I pictures of one kind each)
OUT=$(pwd)/gpurun_out/ikpmc; mkdir -p $OUT; export TMPDIR=/tmp
R=$(pwd)
B="python $R/bench.py --no-cpu-baseline --no-verify --no-other-configs --no-host-packets --no-same-input --variants 1 --gop I --steps 1 --warmup 0"
cd /tmp
for K in 16 4 8; do
  E264_I_KINDS=$K timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM -d $OUT/k$K -- $B > /dev/null 2> $OUT/k$K.err
  echo "== kinds $K"; python $R/tools/pmc_summary.py $(find $OUT/k$K -name '*.db') 2>&1 | grep -A9 intra_kernel
done
cd $R; find $OUT -name '*.db' -size +5M -delete
