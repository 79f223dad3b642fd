# intra kernel in P pictures: time against the share of intra macroblocks (intercept = scan + launch, slope = per macroblock)
mkdir -p gpurun_out/sweep
B="python bench.py --no-cpu-baseline --no-other-configs --no-host-packets --no-same-input --no-verify --variants 1 --steps 6 --warmup 2"
i=20
while read -r KW; do
  i=$((i+1))
  E264_SYNTH_KW="$KW" timeout 300 $B > gpurun_out/sweep/s$i.json 2> gpurun_out/sweep/s$i.err
  python -c "
import json,sys; d=json.load(open('gpurun_out/sweep/s$i.json')); print('%-42s' % sys.argv[1], round(d['value']), {k.split('_')[1]:round(v['ms_per_launch'],3) for k,v in d['roofline']['kernels'].items()})" "$KW"
done <<'KWS'
{"intra_in_inter": 0.0}
{"intra_in_inter": 0.002}
{"intra_in_inter": 0.01}
{"intra_in_inter": 0.025}
{"intra_in_inter": 0.05}
{"intra_in_inter": 0.1}
{"intra_in_inter": 0.2}
KWS
