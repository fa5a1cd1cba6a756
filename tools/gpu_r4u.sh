#!/bin/bash
# round 4: kernel statistics of the configs[4] model's e4m3 step (single stream)
O=$PWD/gpurun_out/r4u; mkdir -p $O; R=$PWD
cd /tmp && export TMPDIR=/tmp
SC_OVERLAP=0 timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o s -- python3 $R/bench.py --model ViT-L-14-genetr --loss spatial --dtype fp8 --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-events --no-loss-delta > $O/prof.log 2>&1
cd $R; find $O -name "*kernel_trace.csv" -delete
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r4u/prof/**/s_kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:22]:
    print(f"{int(r['Calls'])/5:7.1f} {float(r['TotalDurationNs'])/5e6:8.2f} ms/step {float(r['AverageNs'])/1e3:8.1f} us  {r['Name'][:100]}")
print('total', sum(float(r['TotalDurationNs']) for r in rows)/5e6)
PY
