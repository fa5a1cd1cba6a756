#!/bin/bash
O=$PWD/gpurun_out/r3q; mkdir -p $O; R=$PWD
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -4 $O/tests.log
timeout -k 10 120 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log
CASES=60 timeout -k 10 300 python tools/attn_fuzz.py > $O/attn_fuzz.txt 2>&1; tail -3 $O/attn_fuzz.txt
cd /tmp && export TMPDIR=/tmp
SC_OVERLAP=0 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/pmc_mfma -o m -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events --no-loss-delta > $O/pmc_mfma.log 2>&1
cd $R; wc -l $O/pmc_mfma/m_counter_collection.csv
python tools/pmc_generic.py $O/pmc_mfma_summary.json $O/pmc_mfma/m_counter_collection.csv > /dev/null 2>&1; ls -la $O/pmc_mfma_summary.json
find $O -name "*kernel_trace.csv" -size +20M -delete
