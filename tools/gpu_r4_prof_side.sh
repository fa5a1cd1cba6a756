#!/bin/bash
# the step with the weight gradients PINNED to the side stream (SC_OVERLAP=1), kernel statistics
O=$PWD/gpurun_out/r4p2; mkdir -p $O; R=$PWD
cd /tmp && export TMPDIR=/tmp
SC_OVERLAP=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_side -o s -- python3 $R/bench.py --no-cpu-baseline --no-kernel-events --no-loss-delta --steps 6 --warmup 2 > $O/prof_side.log 2>&1
cd $R; find $O -name "*kernel_trace.csv" -delete; find $O -name "s_kernel_stats.csv"
