#!/bin/bash
# round-3 GPU session D: fp8 delayed-scaling path (tests + ViT-L bf16 vs fp8), B=1024 tests, headline bench
O=gpurun_out/r3d; mkdir -p $O
last_json() { python -c "
import json,sys
l=[x for x in open(sys.argv[1]) if x.startswith('{')]
d=json.loads(l[-1]); r=d.get('roofline') or {}
print(sys.argv[2], d['ms_per_step'], d['value'], r.get('achieved'), (r.get('wgrad_tn') or {}).get('achieved'), (r.get('forward_fp8') or {}).get('achieved'), d.get('loss_delta_vs_oracle'))" $1 "$2"; }
timeout -k 10 400 python -m pytest tests/test_gpu_fp8.py tests/test_gpu_gemm.py -x -q -s > $O/tests1.log 2>&1; echo "rc=$?" >> $O/tests1.log; grep -E "^\[fp8|passed|failed|rc=" $O/tests1.log | tail -8
timeout -k 10 600 python -m pytest tests/test_gpu_fullsize.py -x -q -k "configs4" -s > $O/tests2.log 2>&1; echo "rc=$?" >> $O/tests2.log; grep -E "configs4|passed|failed|rc=" $O/tests2.log | tail -8
for dt in bf16 fp8; do
timeout -k 10 240 python bench.py --model ViT-L-14-genetr --dtype $dt --loss spatial --steps 6 --warmup 2 --no-cpu-baseline --no-loss-delta > $O/bench_vitl_$dt.json 2> $O/bench_vitl_$dt.err; last_json $O/bench_vitl_$dt.json "ViT-L $dt"
done
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-loss-delta > $O/bench_vitb.json 2> $O/bench_vitb.err; last_json $O/bench_vitb.json "ViT-B"
timeout -k 10 300 python -m pytest tests/test_gpu_model.py tests/test_gpu_ops.py tests/test_gpu_pipeline.py -x -q > $O/tests3.log 2>&1; echo "rc=$?" >> $O/tests3.log; tail -3 $O/tests3.log
