#!/bin/bash
O=$PWD/gpurun_out/r3ab; mkdir -p $O
SC_HIP_LIB=$PWD/spatial-clip_amd/lib/alt/libspatialclip_hip.so timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "attn or attention" > $O/attn_tests_alt.txt 2>&1; tail -3 $O/attn_tests_alt.txt
bash tools/gpu_r3aa.sh
