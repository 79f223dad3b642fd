M="./edge264_amd/e264_multi --front tools/hostprof/libedge264_front_prof.so --hip edge264_amd/libedge264_hip.so"
S="tests/golden/streams/hd1080_ipp30.264 tests/golden/streams/cabac_hd1080_ibbp30.264"
agg() { python -c "
import sys,re,collections
c=collections.Counter(); n=collections.Counter()
for l in sys.stdin:
    m=re.match(r'emit-profile (.{20})\s+calls\s+(\d+)\s+cycles\s+(\d+)',l)
    if m: c[m.group(1).strip()]+=int(m.group(3)); n[m.group(1).strip()]+=int(m.group(2))
for k in c: print(f'   {k:22s} calls {n[k]:10d}  Mcycles {c[k]/1e6:10.1f}  per call {c[k]/max(n[k],1):8.0f}')
"; }
echo "== parse-only 8 threads (profiled front end)"; timeout 200 $M --threads 8 --repeat 64 --loops 4 --parse-only $S 2> /tmp/po.err | grep -o '"threads.*'; agg < /tmp/po.err
echo "== e2e pinned 8 threads (profiled front end)"; timeout 200 $M --threads 8 --repeat 64 --loops 4 --no-download $S 2> /tmp/e2e.err | grep -o '"threads.*'; agg < /tmp/e2e.err
