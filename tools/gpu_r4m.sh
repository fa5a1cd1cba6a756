#!/bin/bash
# round 4: configs[4] model, weight-gradient grouping modes, bf16 and e4m3
O=$PWD/gpurun_out/r4m; mkdir -p $O
val() { python -c "
import json,sys
l=[x for x in open(sys.argv[1]) if x.startswith('{')]
d=json.loads(l[-1]); print(sys.argv[2], d['ms_per_step'], d['value'])" $1 "$2"; }
for rep in 1 2; do
 for mode in 0 1 3; do
  SC_WGRAD_GROUP=$mode timeout -k 10 400 python bench.py --model ViT-L-14-genetr --loss spatial --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-events --no-loss-delta > $O/bf16_${mode}_$rep.json 2> $O/bf16_${mode}_$rep.err; val $O/bf16_${mode}_$rep.json "bf16 group=$mode"
  SC_WGRAD_GROUP=$mode timeout -k 10 400 python bench.py --model ViT-L-14-genetr --loss spatial --dtype fp8 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-events --no-loss-delta > $O/fp8_${mode}_$rep.json 2> $O/fp8_${mode}_$rep.err; val $O/fp8_${mode}_$rep.json "fp8  group=$mode"
 done
done
