#!/bin/bash
O=gpurun_out/r3g; mkdir -p $O
timeout -k 10 120 python tools/leak_probe.py > $O/leak.txt 2>&1; cat $O/leak.txt | tail -40
timeout -k 10 300 python -m pytest tests/test_gpu_pipeline.py -x -q > $O/tests_pipeline.log 2>&1; echo "rc=$?" >> $O/tests_pipeline.log; tail -5 $O/tests_pipeline.log
