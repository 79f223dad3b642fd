#!/bin/bash
# phase profile of e264_pred_kernel: make -C edge264_amd/csrc variant NAME=phase DEFS=-DE264_PHASE_TIMING
TAG=${1:-ph}; shift
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export E264_ALLOW_ABLATION=1 E264_HIP_LIB=${E264_HIP_LIB:-$REPO/edge264_amd/variants/libedge264_hip_phase.so}
timeout 300 python - "$@" > $OUT/phase.txt 2>$OUT/phase.err <<'PY'
import ctypes as C, os, sys
sys.argv = ["bench.py", "--no-cpu-baseline", "--no-verify", "--steps", "2", "--warmup", "1"] + sys.argv[1:]
sys.path.insert(0, os.getcwd())
import bench
bench.main()
out = (C.c_ulonglong * 160)()
from edge264_amd import backend
L = backend.load_library()
L.e264_debug_phase_cycles.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
assert L.e264_debug_phase_cycles(out, 0) == 0
names = ["setup", "barrier + early-out vote", "classify list 0", "barrier", "items list 0", "barrier", "list 1 (all of it)", "residual lists", "barrier", "residual items", "barrier", "flush"]
tot = sum(out[:12]) or 1
for n, v in zip(names, out[:12]):
    print(f"{n:40s} {v:16d} {100.0 * v / tot:6.2f}%", file=sys.stderr)
import os
dn = ["scan + header", "wait for the row above", "recon: neighbour issue etc", "slice cache + coef commit", "residual", "neighbour commit", "luma prediction steps", "chroma prediction", "store", "release fence", "-", "-", "-", "-"] if os.environ.get("E264_PHASE_INTRA") else ["plan + flush of final groups", "wait for the wave above + top fetch", "commits, prefetch issue, sync", "parameters", "V phase", "H phase", "fence + publish", "-", "-", "-", "-", "-", "-", "-"]
tl = [out[32 + i] for i in range(128)]
t0 = min(v for v in tl if v) if any(tl) else 0
print("deblock timeline of workgroup 0 (us, 100 MHz ticks): group of rows -> start, end", file=sys.stderr)
for q in range(64):
    if tl[2 * q]:
        print(f"  q{q:2d} {(tl[2 * q] - t0) / 100.0:9.1f} {(tl[2 * q + 1] - t0) / 100.0:9.1f}", file=sys.stderr)
tot = sum(out[16:30]) or 1
print("e264_deblock_kernel", file=sys.stderr)
for n, v in zip(dn, out[16:30]):
    print(f"{n:60s} {v:16d} {100.0 * v / tot:6.2f}%", file=sys.stderr)
PY
grep -v "^$" $OUT/phase.err | tail -60
python - <<'PY' >> /dev/null
PY
