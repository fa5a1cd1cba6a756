#!/bin/bash
# same-box A/B of the round-2 tree (.ab/r2, commit 74b8ce3) against the round's final tree, three interleaved pairs
O=$PWD/gpurun_out/r3ai; mkdir -p $O; R=$PWD
val() { python -c "
import json,sys
l=[x for x in open(sys.argv[1]) if x.startswith('{')]
d=json.loads(l[-1]); print(sys.argv[2], d['ms_per_step'], d['value'])" $1 "$2"; }
for rep in 1 2 3; do
  (cd $R/.ab/r2 && timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-loss-delta --no-kernel-events > $O/r2_$rep.json 2> $O/r2_$rep.err); val $O/r2_$rep.json "round-2 tree"
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-loss-delta --no-kernel-events > $O/r3_$rep.json 2> $O/r3_$rep.err; val $O/r3_$rep.json "round-3 tree"
  timeout -k 10 200 python bench.py --residual-stream bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-loss-delta --no-kernel-events > $O/r3b_$rep.json 2> $O/r3b_$rep.err; val $O/r3b_$rep.json "round-3 tree, bf16 residual stream"
done
