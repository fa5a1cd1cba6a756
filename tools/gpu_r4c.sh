#!/bin/bash
# round 4: head autograd bridge without ATen ops; loss / model / ddp tests, aten call sites, single-stream kernel stats
O=$PWD/gpurun_out/r4c; mkdir -p $O; R=$PWD
timeout -k 10 900 python -m pytest tests/test_gpu_loss.py tests/test_gpu_model.py tests/test_gpu_ddp.py tests/test_gpu_zero_shot.py -x -q -m gpu > $O/tests.txt 2>&1; rc=$?; tail -5 $O/tests.txt
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python tools/aten_sites.py > $O/aten_sites.txt 2>&1; tail -8 $O/aten_sites.txt
cd /tmp && export TMPDIR=/tmp
SC_OVERLAP=0 timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o s -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-events --no-loss-delta > $O/prof.log 2>&1
cd $R
find $O -name "*kernel_trace.csv" -delete
find $O -name "*kernel_stats.csv" | head; tail -c 400 $O/prof.log
