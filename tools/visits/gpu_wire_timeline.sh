#!/bin/bash
# timeline of the wire leg of tools/wire_probe.py: kernel and copy intervals (rocprofv3 --kernel-trace --memory-copy-trace, no counters), the idle gaps between
# consecutive kernels of the last 40 batches and the H2D copies beside them.  usage: tools/visits/gpu_wire_timeline.sh TAG [file]
TAG=${1:-wiretl}; OUT=gpurun_out/$TAG; mkdir -p $OUT
F=${2:-nat1080_ipp30.264}
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/wtl -- python $GRAFT_REPO_ROOT/tools/wire_probe.py $F 256 > $GRAFT_REPO_ROOT/$OUT/probe.json 2> $GRAFT_REPO_ROOT/$OUT/probe.err
cd $GRAFT_REPO_ROOT
python - <<PY > $OUT/timeline.txt
import csv, glob
k = glob.glob('/tmp/wtl/**/*kernel_trace.csv', recursive=True)[0]
m = glob.glob('/tmp/wtl/**/*memory_copy_trace.csv', recursive=True)[0]
K = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:28]) for r in csv.DictReader(open(k))]
M = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Direction'], int(r.get('Bytes', 0) or 0)) for r in csv.DictReader(open(m))]
K.sort(); M.sort()
tail = K[-5 * 40:]
t0 = tail[0][0]
print('last 40 batches of the wire leg: kernel start, duration, gap before it (us)')
prev = None
gaps = {}
for s, e, n in tail:
    g = (s - prev) / 1e3 if prev else 0.0
    gaps.setdefault(n, []).append(g)
    prev = e
for n, g in gaps.items():
    print(f'  gap before {n:30s} mean {sum(g)/len(g):8.1f} us  max {max(g):8.1f}')
dur = {}
for s, e, n in tail:
    dur.setdefault(n, []).append((e - s) / 1e3)
for n, d in dur.items():
    print(f'  duration   {n:30s} mean {sum(d)/len(d):8.1f} us')
print('span per batch (us):', (tail[-1][1] - tail[0][0]) / 1e3 / 40)
big = [(s, e, d, b) for s, e, d, b in M if s >= t0 and b > 1 << 20]
if big:
    print('large copies in that window:', len(big), 'mean ms', sum(e - s for s, e, d, b in big) / 1e6 / len(big), 'mean MB', sum(b for *_, b in big) / 1e6 / len(big), 'GB/s', sum(b for *_, b in big) / sum(e - s for s, e, d, b in big))
for s, e, n in tail[:15]:
    print(f'  {(s - t0) / 1e3:10.1f} {(e - s) / 1e3:8.1f}  {n}')
for s, e, d, b in [x for x in M if x[0] >= t0][:12]:
    print(f'  copy {(s - t0) / 1e3:10.1f} {(e - s) / 1e3:8.1f} {d} {b}')
PY
cat $OUT/timeline.txt
