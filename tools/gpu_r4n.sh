#!/bin/bash
# round 4: e4m3 weight-gradient kernel: exactness, then time against the bf16 TN kernel on the ViT-L/14 MLP shapes
O=$PWD/gpurun_out/r4n; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_fp8.py -x -q -m gpu -k "wgrad_fp8" > $O/tests.txt 2>&1; rc=$?; tail -12 $O/tests.txt
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python tools/bench_wgrad_fp8.py > $O/bench.txt 2>&1; cat $O/bench.txt
