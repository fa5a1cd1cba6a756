#!/bin/bash
# round 4: grouped weight-gradient launches: tests, then same-box A/B of SC_WGRAD_GROUP = 0 / 2 / 4, three interleaved rounds
O=$PWD/gpurun_out/r4f; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_gemm.py -x -q -m gpu -k group > $O/tests.txt 2>&1; rc=$?; tail -5 $O/tests.txt
[ $rc -ne 0 ] && exit $rc
val() { python -c "
import json,sys
l=[x for x in open(sys.argv[1]) if x.startswith('{')]
d=json.loads(l[-1]); print(sys.argv[2], d['ms_per_step'], d['value'], d.get('loss_delta_vs_oracle'))" $1 "$2"; }
for rep in 1 2 3; do
  for mode in 0 1 3; do
    SC_WGRAD_GROUP=$mode timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-loss-delta --no-kernel-events > $O/m${mode}_$rep.json 2> $O/m${mode}_$rep.err; val $O/m${mode}_$rep.json "SC_WGRAD_GROUP=$mode"
  done
done
for mode in 0 1 3; do
  SC_OVERLAP=0 SC_WGRAD_GROUP=$mode timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-loss-delta --no-kernel-events > $O/s${mode}.json 2> $O/s${mode}.err; val $O/s${mode}.json "single stream, SC_WGRAD_GROUP=$mode"
done
