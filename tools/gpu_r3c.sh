#!/bin/bash
# round-3 GPU session C: short-tile GEMM tests + A/B, fp8 + configs[4] B=1024 tests, headline and ViT-L benches
O=gpurun_out/r3c; mkdir -p $O
last_json() { python -c "
import json,sys
l=[x for x in open(sys.argv[1]) if x.startswith('{')]
d=json.loads(l[-1]); r=d.get('roofline') or {}
print(sys.argv[2], d['ms_per_step'], d['value'], r.get('achieved'), (r.get('wgrad_tn') or {}).get('achieved'), (r.get('forward_fp8') or {}).get('achieved'))" $1 "$2"; }
timeout -k 10 400 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_fp8.py -x -q > $O/tests1.log 2>&1; echo "rc=$?" >> $O/tests1.log; tail -4 $O/tests1.log
timeout -k 10 200 python tools/bench_short_tiles.py > $O/short_tiles.txt 2>&1; cat $O/short_tiles.txt
timeout -k 10 500 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_model.py -x -q -k "batch_1024 or model or vit or step" -s > $O/tests2.log 2>&1; echo "rc=$?" >> $O/tests2.log; grep -E "configs4|passed|failed|rc=" $O/tests2.log | tail -6
for f in 16 0; do
SC_GEMM_TILE_F=$f timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-loss-delta > $O/bench_vitb_F$f.json 2> $O/bench_vitb_F$f.err; last_json $O/bench_vitb_F$f.json "ViT-B F=$f"
done
for dt in bf16 fp8; do
timeout -k 10 240 python bench.py --model ViT-L-14-genetr --dtype $dt --loss spatial --steps 6 --warmup 2 --no-cpu-baseline --no-loss-delta > $O/bench_vitl_$dt.json 2> $O/bench_vitl_$dt.err; last_json $O/bench_vitl_$dt.json "ViT-L $dt"
done
SC_GEMM_TILE_F=16 timeout -k 10 240 python bench.py --model ViT-L-14-genetr --dtype bf16 --loss spatial --steps 6 --warmup 2 --no-cpu-baseline --no-loss-delta --no-kernel-events > $O/bench_vitl_bf16_F16.json 2> $O/bench_vitl_bf16_F16.err; last_json $O/bench_vitl_bf16_F16.json "ViT-L bf16 F=16"
