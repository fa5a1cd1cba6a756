#!/bin/bash
# round 4: the bench line as the driver runs it, SpatialLoss variant, forced-dist variant, fp32-stream variant; same-box A/B against the round-3 tree
O=$PWD/gpurun_out/r4k; mkdir -p $O; R=$PWD
timeout -k 10 500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_line.err; tail -c 600 $O/bench_line.err; python -c "
import json; d=json.loads([l for l in open('$O/bench_line.json') if l.startswith('{')][-1]); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['whole_step']['frac'], d['loss_delta_vs_oracle'], d['max_abs_feature_delta'], d['roofline'].get('wgrad_tn'))"
val() { python -c "
import json,sys
l=[x for x in open(sys.argv[1]) if x.startswith('{')]
d=json.loads(l[-1]); print(sys.argv[2], d['ms_per_step'], d['value'], d.get('loss_delta_vs_oracle'))" $1 "$2"; }
for rep in 1 2 3; do
  (cd $R/.ab/r3 && timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-loss-delta --no-kernel-events > $O/r3_$rep.json 2> $O/r3_$rep.err); val $O/r3_$rep.json "round-3 tree"
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-loss-delta --no-kernel-events > $O/r4_$rep.json 2> $O/r4_$rep.err; val $O/r4_$rep.json "round-4 tree"
  timeout -k 10 200 python bench.py --residual-stream fp32 --steps 20 --warmup 5 --no-cpu-baseline --no-loss-delta --no-kernel-events > $O/r4f_$rep.json 2> $O/r4f_$rep.err; val $O/r4f_$rep.json "round-4 tree, fp32 residual stream"
done
timeout -k 10 300 python bench.py --loss spatial --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-events > $O/spatial.json 2> $O/spatial.err; val $O/spatial.json "SpatialLoss"
SC_FORCE_DIST=1 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-loss-delta --no-kernel-events > $O/dist.json 2> $O/dist.err; val $O/dist.json "SC_FORCE_DIST=1"
