#!/bin/bash
O=$PWD/gpurun_out/r4z; mkdir -p $O; R=$PWD
cd /tmp && export TMPDIR=/tmp
for ov in 0 1; do
SC_OVERLAP=$ov timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/t$ov -o t -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-events --no-loss-delta > $O/t$ov.log 2>&1
python3 $R/tools/trace_gaps.py $(find $O/t$ov -name "t_kernel_trace.csv") | tee $O/gaps_overlap$ov.txt
done
find $O -name "*kernel_trace.csv" -delete
