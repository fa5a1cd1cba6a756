#!/bin/bash
O=gpurun_out/r4w; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_fp8.py tests/test_gpu_ops.py -x -q -k "e4m3_copies or attn" > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
timeout -k 10 200 python tools/bench_attn257.py 2>&1 | tee $O/attn257.txt
