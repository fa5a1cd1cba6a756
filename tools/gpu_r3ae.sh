#!/bin/bash
# residual stream fp32 vs bf16 (SC_RES_STREAM): same-box A/B of the headline, three interleaved pairs, then parity legs
O=$PWD/gpurun_out/r3ae; mkdir -p $O
val() { python -c "
import json,sys
l=[x for x in open(sys.argv[1]) if x.startswith('{')]
d=json.loads(l[-1]); p=d.get('parity') or {}
print(sys.argv[2], d['ms_per_step'], d['value'], {k:(round(v['loss_delta_vs_oracle'],6), round(v['max_abs_feature_delta'],5)) for k,v in p.items()} if p else '')" $1 "$2"; }
for rep in 1 2 3; do
  SC_RES_STREAM=fp32 timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-loss-delta --no-kernel-events > $O/f32_$rep.json 2> $O/f32_$rep.err; val $O/f32_$rep.json "fp32 stream"
  SC_RES_STREAM=bf16 timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-loss-delta --no-kernel-events > $O/b16_$rep.json 2> $O/b16_$rep.err; val $O/b16_$rep.json "bf16 stream"
done
for m in fp32 bf16; do
  SC_RES_STREAM=$m timeout -k 10 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events > $O/parity_$m.json 2> $O/parity_$m.err; val $O/parity_$m.json "parity $m"
  SC_RES_STREAM=$m timeout -k 10 400 python bench.py --loss spatial --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events > $O/parity_spatial_$m.json 2> $O/parity_spatial_$m.err; val $O/parity_spatial_$m.json "parity spatial $m"
done
