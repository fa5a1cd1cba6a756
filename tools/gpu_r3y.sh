#!/bin/bash
# bf16 residual-gradient stream: model tests, then same-box A/B of the headline (SC_RES_GRAD=fp32 vs bf16), three interleaved pairs
O=$PWD/gpurun_out/r3y; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_ops.py tests/test_gpu_fp8.py tests/test_gpu_fullsize.py -x -q -m gpu > $O/tests.txt 2>&1; tail -6 $O/tests.txt
val() { python -c "
import json,sys
l=[x for x in open(sys.argv[1]) if x.startswith('{')]
d=json.loads(l[-1]); print(sys.argv[2], d['ms_per_step'], d['value'], d.get('loss_delta_vs_oracle'))" $1 "$2"; }
for rep in 1 2 3; do
  SC_RES_GRAD=fp32 timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-loss-delta --no-kernel-events > $O/f32_$rep.json 2> $O/f32_$rep.err; val $O/f32_$rep.json "fp32 residual gradient"
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-loss-delta --no-kernel-events > $O/b16_$rep.json 2> $O/b16_$rep.err; val $O/b16_$rep.json "bf16 residual gradient"
done
