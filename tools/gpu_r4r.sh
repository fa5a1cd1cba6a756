#!/bin/bash
# round 4: configs[4] at its stated 1024 pairs per GPU with activation recomputation: property tests, then bf16 vs e4m3
O=$PWD/gpurun_out/r4r; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "1024" -s > $O/tests.txt 2>&1; rc=$?; grep -E "configs4|passed|failed|Error" $O/tests.txt | tail -8
[ $rc -ne 0 ] && exit $rc
val() { python -c "
import json,sys
l=[x for x in open(sys.argv[1]) if x.startswith('{')]
d=json.loads(l[-1]); print(sys.argv[2], d['ms_per_step'], d['value'], d.get('hbm_peak_gib'))" $1 "$2"; }
timeout -k 10 500 python bench.py --model ViT-L-14-genetr --loss spatial --batch 1024 --grad-checkpointing --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-events --no-loss-delta > $O/bf16.json 2> $O/bf16.err; val $O/bf16.json "bf16 B=1024 recompute"
timeout -k 10 500 python bench.py --model ViT-L-14-genetr --loss spatial --batch 1024 --grad-checkpointing --dtype fp8 --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-events --no-loss-delta > $O/fp8.json 2> $O/fp8.err; val $O/fp8.json "e4m3 B=1024 recompute"
