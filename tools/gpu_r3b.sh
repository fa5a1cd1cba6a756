#!/bin/bash
# round-3 GPU session B: fused-quantiser tests, configs[4] at B=1024, headline bench, ViT-L bf16 vs fp8, forced-dist line
O=gpurun_out/r3b; mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_gpu_fp8.py tests/test_gpu_ops.py tests/test_gpu_model.py -x -q > $O/tests1.log 2>&1; echo "rc=$?" >> $O/tests1.log; tail -3 $O/tests1.log
timeout -k 10 400 python -m pytest tests/test_gpu_fullsize.py -x -q -k "configs4" -s > $O/tests2.log 2>&1; echo "rc=$?" >> $O/tests2.log; grep -E "configs4|passed|failed|rc=" $O/tests2.log | tail -8
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-loss-delta > $O/bench_vitb.json 2> $O/bench_vitb.err; python -c "
import json;d=json.load(open('$O/bench_vitb.json'));print('ViT-B',d['ms_per_step'],d['value'],d['roofline']['achieved'],d['roofline']['wgrad_tn'])"
for dt in bf16 fp8; do
timeout -k 10 240 python bench.py --model ViT-L-14-genetr --dtype $dt --loss spatial --steps 6 --warmup 2 --no-cpu-baseline --no-loss-delta > $O/bench_vitl_$dt.json 2> $O/bench_vitl_$dt.err; python -c "
import json;d=json.load(open('$O/bench_vitl_$dt.json'));print('ViT-L $dt',d['ms_per_step'],d['value'],d['roofline'].get('forward_fp8'),d['roofline']['achieved'])"
done
SC_FORCE_DIST=1 timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-loss-delta --no-kernel-events > $O/bench_forcedist.json 2> $O/bench_forcedist.err; python -c "
import json;d=json.load(open('$O/bench_forcedist.json'));print('forced dist',d['ms_per_step'],d['value'])"
