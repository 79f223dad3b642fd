#!/bin/bash
# the RCCL code path of bench.py on one GPU (torchrun, 1 rank, process group forced)
export TMPDIR=/tmp
mkdir -p gpurun_out/dist1
E264_FORCE_DIST=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/dist1/out.json 2> gpurun_out/dist1/err.log
echo rc=$?; cat gpurun_out/dist1/out.json | cut -c1-400; tail -5 gpurun_out/dist1/err.log
