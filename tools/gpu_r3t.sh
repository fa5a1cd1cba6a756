#!/bin/bash
# PMC passes on the PNG decode kernel at 8192 and 256 tiles per launch (instruction mix, busy cycles)
O=$PWD/gpurun_out/r3t; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for N in 8192 256; do
  timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES --kernel-trace --output-format csv -d $O/p1_$N -o x -- python3 $R/tools/png_pmc_target.py $N > $O/p1_$N.log 2>&1 || exit 1
  timeout -k 10 200 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $O/p2_$N -o x -- python3 $R/tools/png_pmc_target.py $N > $O/p2_$N.log 2>&1 || exit 1
  timeout -k 10 200 rocprofv3 --pmc SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_BRANCH --kernel-trace --output-format csv -d $O/p3_$N -o x -- python3 $R/tools/png_pmc_target.py $N > $O/p3_$N.log 2>&1 || exit 1
  python3 $R/tools/pmc_generic.py $O/pmc_$N.json "$O/p1_$N/**/x_counter_collection.csv" "$O/p2_$N/**/x_counter_collection.csv" "$O/p3_$N/**/x_counter_collection.csv" | grep -i png
done
