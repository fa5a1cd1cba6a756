#!/bin/bash
# pipeline tests + end-to-end training from shards_v1
O=$PWD/gpurun_out/r3u; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_gpu_pipeline.py -x -q -m gpu > $O/pipe_tests.txt 2>&1; tail -5 $O/pipe_tests.txt
timeout -k 10 600 python tools/bench_shards_training.py > $O/shards_training.txt 2>&1; grep -v amdgpu $O/shards_training.txt | tail -15
