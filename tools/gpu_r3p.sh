#!/bin/bash
# round-3 profile session: bench lines + rocprofv3 kernel stats (single stream / shipped schedule) + PMC passes
O=$PWD/gpurun_out/r3p; mkdir -p $O
R=$PWD
python bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_line.err; tail -c 600 $O/bench_line.json
python bench.py --loss spatial --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events > $O/bench_line_spatial.json 2> $O/bench_spatial.err
SC_FORCE_DIST=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-loss-delta --no-kernel-events > $O/bench_line_forced_dist_one_rank.json 2> $O/bench_fd.err
cd /tmp && export TMPDIR=/tmp
SC_OVERLAP=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_single -o s -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-events --no-loss-delta > $O/prof_single.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_side -o s -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-events --no-loss-delta > $O/prof_side.log 2>&1
SC_OVERLAP=0 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mfma -o m -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events --no-loss-delta > $O/pmc_mfma.log 2>&1
SC_OVERLAP=0 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f -o f -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-events --no-loss-delta > $O/pmc_f.log 2>&1
SC_OVERLAP=0 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w -o w -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-events --no-loss-delta > $O/pmc_w.log 2>&1
cd $R
find $O -name "*.csv" | head -20
# keep the merge small: stats + counter collections only
find $O -name "*kernel_trace.csv" -size +20M -delete
python tools/pmc_summary.py $(find $O/pmc_f -name "*counter_collection.csv") $(find $O/pmc_w -name "*counter_collection.csv") $O/pmc_traffic_summary.json > /dev/null 2>&1
ls -la $O | head -30
