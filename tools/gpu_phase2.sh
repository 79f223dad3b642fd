#!/bin/bash
# phase profile of e264_pred_kernel: make -C edge264_amd/csrc variant NAME=phase DEFS=-DE264_PHASE_TIMING
TAG=${1:-ph}; shift
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export E264_HIP_LIB=$REPO/edge264_amd/variants/libedge264_hip_phase.so
timeout 300 python - "$@" > $OUT/phase.txt 2>$OUT/phase.err <<'PY'
import ctypes as C, os, sys
sys.argv = ["bench.py", "--no-cpu-baseline", "--no-verify", "--steps", "2", "--warmup", "1"] + sys.argv[1:]
sys.path.insert(0, os.getcwd())
import bench
bench.main()
out = (C.c_ulonglong * 32)()
from edge264_amd import backend
L = backend.load_library()
L.e264_debug_phase_cycles.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
assert L.e264_debug_phase_cycles(out, 0) == 0
names = ["setup", "barrier + early-out vote", "classify list 0", "barrier", "items list 0", "barrier", "list 1 (all of it)", "residual lists", "barrier", "residual items", "barrier", "flush"]
tot = sum(out[:12])
for n, v in zip(names, out[:12]):
    print(f"{n:40s} {v:16d} {100.0 * v / tot:6.2f}%", file=sys.stderr)
PY
grep -v "^$" $OUT/phase.err | tail -14
