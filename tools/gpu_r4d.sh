#!/bin/bash
# round 4: attention backward with the dS ring (sc_attention_bwd3.hip): op tests, then the kernels alone
O=$PWD/gpurun_out/r4d; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "attention" > $O/tests.txt 2>&1; rc=$?; tail -5 $O/tests.txt
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python tools/bench_attn.py > $O/bench_attn.txt 2>&1; cat $O/bench_attn.txt
