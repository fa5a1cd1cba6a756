#!/bin/bash
O=$PWD/gpurun_out/r4q; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_fp8.py -x -q -m gpu -k "e4m3_mlp_weight or delayed_scaling_state" > $O/tests.txt 2>&1; rc=$?; tail -15 $O/tests.txt
exit $rc
