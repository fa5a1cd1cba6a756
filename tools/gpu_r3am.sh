#!/bin/bash
# configs[4] model (ViT-L/14 + gene transformer, B = 256, bf16): kernel stats, single stream
O=$PWD/gpurun_out/r3am; mkdir -p $O; R=$PWD
cd /tmp && export TMPDIR=/tmp
SC_OVERLAP=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o s -- python3 $R/bench.py --model ViT-L-14-genetr --loss spatial --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-events --no-loss-delta > $O/prof.log 2>&1
cd $R
find $O -name "*kernel_trace.csv" -delete
tail -c 300 $O/prof.log
