#!/bin/bash
# round 4: attention backward ring kernel in the training step, same-box A/B (SC_ATTN_BWD3=0 -> round-3 single-pass kernel)
O=$PWD/gpurun_out/r4e; mkdir -p $O
val() { python -c "
import json,sys
l=[x for x in open(sys.argv[1]) if x.startswith('{')]
d=json.loads(l[-1]); print(sys.argv[2], d['ms_per_step'], d['value'], d.get('loss_delta_vs_oracle'))" $1 "$2"; }
for rep in 1 2 3; do
  SC_ATTN_BWD3=0 timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-loss-delta --no-kernel-events > $O/old_$rep.json 2> $O/old_$rep.err; val $O/old_$rep.json "single-pass (round 3) attention backward"
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-loss-delta --no-kernel-events > $O/new_$rep.json 2> $O/new_$rep.err; val $O/new_$rep.json "ring attention backward"
done
REPS=4 timeout -k 10 300 python tools/bench_attn.py > $O/bench_attn.txt 2>&1; grep -v amdgpu $O/bench_attn.txt
timeout -k 10 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_fullsize.py -x -q -m gpu > $O/tests.txt 2>&1; tail -3 $O/tests.txt
