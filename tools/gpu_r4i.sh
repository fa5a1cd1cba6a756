#!/bin/bash
# round 4, verdict item 7: where do the NT GEMMs' 1.5x fetched bytes come from?  L2 hit / miss and fabric read requests per kernel.
O=$PWD/gpurun_out/r4i; mkdir -p $O; R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/counters.txt 2>&1
grep -o "TCC_[A-Z0-9_]*\(sum\)\?" $O/counters.txt | sort -u | tr '\n' ' ' | head -c 3000; echo
run() { # name, counters...
  name=$1; shift
  SC_OVERLAP=0 timeout -k 10 400 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$name -o x -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events --no-loss-delta > $O/$name.log 2>&1
  echo "$name rc=$?"
}
run l2a TCC_HIT_sum TCC_MISS_sum
run l2b TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
run l2c TCC_REQ_sum TCC_READ_sum
cd $R
python tools/pmc_generic.py $O/pmc_l2.json "$O/l2a/**/x_counter_collection.csv" "$O/l2b/**/x_counter_collection.csv" "$O/l2c/**/x_counter_collection.csv" > $O/summary.txt 2>&1; head -30 $O/summary.txt
find $O -name "*kernel_trace.csv" -delete; find $O -name "x_counter_collection.csv" -size +20M -delete
