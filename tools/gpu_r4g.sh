#!/bin/bash
# round 4: LayerNorm forward with two bf16 rows per wave: op tests, kernel alone (old / new), step A/B
O=$PWD/gpurun_out/r4g; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "layernorm or norm" > $O/tests.txt 2>&1; rc=$?; tail -3 $O/tests.txt
[ $rc -ne 0 ] && exit $rc
SC_LN_FWD2=0 timeout -k 10 200 python tools/bench_ln.py > $O/ln_old.txt 2>&1; grep "ln_fwd" $O/ln_old.txt
timeout -k 10 200 python tools/bench_ln.py > $O/ln_new.txt 2>&1; grep "ln_" $O/ln_new.txt
val() { python -c "
import json,sys
l=[x for x in open(sys.argv[1]) if x.startswith('{')]
d=json.loads(l[-1]); print(sys.argv[2], d['ms_per_step'], d['value'], d.get('loss_delta_vs_oracle'))" $1 "$2"; }
for rep in 1 2 3; do
  SC_LN_FWD2=0 timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-loss-delta --no-kernel-events > $O/old_$rep.json 2> $O/old_$rep.err; val $O/old_$rep.json "one row per wave"
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-loss-delta --no-kernel-events > $O/new_$rep.json 2> $O/new_$rep.err; val $O/new_$rep.json "two rows per wave"
done
