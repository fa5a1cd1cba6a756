#!/bin/bash
# round 4: configs[4] model (ViT-L/14 + gene transformer, B = 256): bf16 vs e4m3, interleaved twice; kernel stats of the bf16 step
O=$PWD/gpurun_out/r4l; mkdir -p $O; R=$PWD
val() { python -c "
import json,sys
l=[x for x in open(sys.argv[1]) if x.startswith('{')]
d=json.loads(l[-1]); print(sys.argv[2], d['ms_per_step'], d['value'], d.get('loss_delta_vs_oracle'), d.get('max_abs_feature_delta'))" $1 "$2"; }
for rep in 1 2; do
  timeout -k 10 400 python bench.py --model ViT-L-14-genetr --loss spatial --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-events --no-loss-delta > $O/bf16_$rep.json 2> $O/bf16_$rep.err; val $O/bf16_$rep.json "bf16"
  timeout -k 10 400 python bench.py --model ViT-L-14-genetr --loss spatial --dtype fp8 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-events --no-loss-delta > $O/fp8_$rep.json 2> $O/fp8_$rep.err; val $O/fp8_$rep.json "fp8"
done
cd /tmp && export TMPDIR=/tmp
SC_OVERLAP=0 timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o s -- python3 $R/bench.py --model ViT-L-14-genetr --loss spatial --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-events --no-loss-delta > $O/prof.log 2>&1
cd $R; find $O -name "*kernel_trace.csv" -delete
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r4l/prof/**/s_kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:16]:
    print(f"{int(r['Calls'])/4:7.1f} {float(r['TotalDurationNs'])/4e6:8.2f} ms/step {float(r['AverageNs'])/1e3:8.1f} us  {r['Name'][:90]}")
print('total', sum(float(r['TotalDurationNs']) for r in rows)/4e6)
PY
