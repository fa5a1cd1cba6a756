#!/bin/bash
# round 4, steps 1 + 4: gelu'(u) stored by the forward epilogue; bf16 residual stream as the default.  Whole GPU suite
# (all failures listed), then same-box A/B of the round-3 tree (.ab/r3) against this tree, three interleaved pairs.
O=$PWD/gpurun_out/r4a; mkdir -p $O; R=$PWD
timeout -k 10 1000 python -m pytest tests -q -m gpu > $O/tests.txt 2>&1; rc=$?; grep -E "^(FAILED|ERROR)|passed|failed" $O/tests.txt | tail -30
val() { python -c "
import json,sys
l=[x for x in open(sys.argv[1]) if x.startswith('{')]
d=json.loads(l[-1]); print(sys.argv[2], d['ms_per_step'], d['value'], d.get('loss_delta_vs_oracle'))" $1 "$2"; }
for rep in 1 2 3; do
  (cd $R/.ab/r3 && timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-loss-delta --no-kernel-events > $O/r3_$rep.json 2> $O/r3_$rep.err); val $O/r3_$rep.json "round-3 tree"
  SC_RES_STREAM=fp32 timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-loss-delta --no-kernel-events > $O/r4f_$rep.json 2> $O/r4f_$rep.err; val $O/r4f_$rep.json "round-4 tree, fp32 stream"
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-loss-delta --no-kernel-events > $O/r4_$rep.json 2> $O/r4_$rep.err; val $O/r4_$rep.json "round-4 tree (bf16 stream)"
done
exit $rc
