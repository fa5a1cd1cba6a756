#!/bin/bash
# round 4: e4m3 MLP weight gradients in the model: fp8 tests + configs[4] parity, then bf16 / e4m3 (SC_FP8_WGRAD=0) / e4m3 interleaved twice
O=$PWD/gpurun_out/r4o; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_fp8.py tests/test_gpu_fullsize.py -x -q -m gpu -k "fp8 or configs4" > $O/tests.txt 2>&1; rc=$?; tail -15 $O/tests.txt; grep -E "configs4|fp8 model" $O/tests.txt | head
[ $rc -ne 0 ] && exit $rc
val() { python -c "
import json,sys
l=[x for x in open(sys.argv[1]) if x.startswith('{')]
d=json.loads(l[-1]); print(sys.argv[2], d['ms_per_step'], d['value'])" $1 "$2"; }
for rep in 1 2; do
  timeout -k 10 400 python bench.py --model ViT-L-14-genetr --loss spatial --steps 5 --warmup 3 --no-cpu-baseline --no-kernel-events --no-loss-delta > $O/bf16_$rep.json 2> $O/bf16_$rep.err; val $O/bf16_$rep.json "bf16"
  SC_FP8_WGRAD=0 timeout -k 10 400 python bench.py --model ViT-L-14-genetr --loss spatial --dtype fp8 --steps 5 --warmup 3 --no-cpu-baseline --no-kernel-events --no-loss-delta > $O/fp8a_$rep.json 2> $O/fp8a_$rep.err; val $O/fp8a_$rep.json "e4m3, bf16 weight gradients (round 3 recipe)"
  timeout -k 10 400 python bench.py --model ViT-L-14-genetr --loss spatial --dtype fp8 --steps 5 --warmup 3 --no-cpu-baseline --no-kernel-events --no-loss-delta > $O/fp8b_$rep.json 2> $O/fp8b_$rep.err; val $O/fp8b_$rep.json "e4m3 + e4m3 MLP weight gradients"
done
