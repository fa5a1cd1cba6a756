#!/bin/bash
# round-3 GPU session F: the whole -m gpu suite, then fp8 delayed-scaling A/B on one box
O=gpurun_out/r3f; mkdir -p $O
last_json() { python -c "
import json,sys
l=[x for x in open(sys.argv[1]) if x.startswith('{')]
d=json.loads(l[-1]); r=d.get('roofline') or {}
print(sys.argv[2], d['ms_per_step'], d['value'], r.get('achieved'), (r.get('wgrad_tn') or {}).get('achieved'), (r.get('forward_fp8') or {}).get('achieved'))" $1 "$2"; }
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -4 $O/tests.log
for v in bf16 fp8d0 fp8d1 bf16 fp8d0 fp8d1; do
  case $v in bf16) dt=bf16; dl=1;; fp8d0) dt=fp8; dl=0;; fp8d1) dt=fp8; dl=1;; esac
  SC_FP8_DELAYED=$dl timeout -k 10 240 python bench.py --model ViT-L-14-genetr --dtype $dt --loss spatial --steps 6 --warmup 2 --no-cpu-baseline --no-loss-delta --no-kernel-events > $O/bench_vitl_$v.json 2> $O/bench_vitl_$v.err; last_json $O/bench_vitl_$v.json "ViT-L $v"
done
