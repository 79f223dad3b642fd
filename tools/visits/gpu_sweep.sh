# what each kernel's time depends on: the bench GOP with one property of the synthetic content changed at a time (E264_SYNTH_KW)
mkdir -p gpurun_out/sweep
B="python bench.py --no-cpu-baseline --no-other-configs --no-host-packets --no-same-input --variants 1 --steps 6 --warmup 2"
i=0
while read -r KW; do
  i=$((i+1))
  E264_SYNTH_KW="$KW" timeout 300 $B > gpurun_out/sweep/s$i.json 2> gpurun_out/sweep/s$i.err
  python -c "
import json,sys; d=json.load(open('gpurun_out/sweep/s$i.json')); print('%-42s' % sys.argv[1], round(d['value']), d['bit_exact'], {k.split('_')[1]:round(v['ms_per_launch'],3) for k,v in d['roofline']['kernels'].items()})" "$KW"
done <<'KWS'
{}
{"mv_range": 0}
{"mv_range": 3}
{"residual_prob": 0.0}
{"residual_prob": 1.0}
{"t8x8": false}
{"intra_in_inter": 0.0}
{"p_skip": 1.0}
{"num_refs": 1}
KWS
