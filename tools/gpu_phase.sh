#!/bin/bash
# Phase profile of e264_mbpar_kernel: build `make -C edge264_amd/csrc variant NAME=phase DEFS=-DE264_PHASE_TIMING` first.
# Runs the default bench workload on the instrumented library and prints the share of wave wall-cycles per phase.
TAG=${1:-phase}; shift
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export E264_HIP_LIB=$REPO/edge264_amd/variants/libedge264_hip_phase.so
timeout 300 python - "$@" > $OUT/phase.txt 2>$OUT/phase.err <<'PY'
import ctypes as C, os, subprocess, sys, json
lib = C.CDLL(os.environ["E264_HIP_LIB"])
# same process: run bench.main() with the instrumented library, then read the counters
sys.argv = ["bench.py", "--no-cpu-baseline", "--no-verify", "--steps", "2", "--warmup", "1"] + sys.argv[1:]
sys.path.insert(0, os.getcwd())
import bench
r, w = os.pipe()
bench.main()
out = (C.c_ulonglong * 32)()
from edge264_amd import backend
L = backend.load_library()
L.e264_debug_phase_cycles.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
assert L.e264_debug_phase_cycles(out, 0) == 0
names = ["prologue", "top: mc_finish (raw motion)", "top: slice cache + coef/window commit (consumes the prefetches)", "wave_sync", "issue: chroma taps (rest of mc_issue)",
         "residual", "mc_compute list 0", "list 1", "residual add + staging + end sync", "strip_flush",
         "issue: coef", "issue: luma windows (up to mc_issue chroma part)", "issue: raw motion", "next header (mb_from_lanes)"]
tot = sum(out[:14])
for n, v in zip(names, out[:14]):
    print(f"{n:70s} {v:16d} {100.0 * v / tot:6.2f}%", file=sys.stderr)
dn = ["loop top", "wait for the row above / ring back-pressure", "top rows from the ring", "prefetch next macroblock (issue)", "carry + tile fill (consumes the prefetch)",
      "vertical edges", "horizontal edges", "ring publish", "flush of the last group (row end)", "fence, progress, cur = nxt (waits for the prefetch)", "stage: carried columns (+ flush of the completed group every 4th step)", "stage: this macroblock", "-", "-"]
tot = sum(out[16:30]) or 1
print("e264_deblock_kernel", file=sys.stderr)
for n, v in zip(dn, out[16:30]):
    print(f"{n:70s} {v:16d} {100.0 * v / tot:6.2f}%", file=sys.stderr)
PY
cat $OUT/phase.err | grep -v "^$" | tail -30
