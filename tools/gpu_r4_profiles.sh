#!/bin/bash
# round-4 profile session (headline model): bench lines, rocprofv3 kernel stats (single stream / shipped schedule), PMC traffic and MFMA busy
O=$PWD/gpurun_out/r4p; mkdir -p $O
R=$PWD
python bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_line.err; tail -c 300 $O/bench_line.err
python bench.py --loss spatial --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events > $O/bench_line_spatial.json 2> $O/bench_spatial.err
python bench.py --residual-stream fp32 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-events > $O/bench_line_res_stream_fp32.json 2> $O/bench_rs.err
SC_FORCE_DIST=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-loss-delta --no-kernel-events > $O/bench_line_forced_dist_one_rank.json 2> $O/bench_fd.err
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-kernel-events --no-loss-delta"
SC_OVERLAP=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_single -o s -- $B --steps 6 --warmup 2 > $O/prof_single.log 2>&1
SC_OVERLAP=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_side -o s -- $B --steps 6 --warmup 2 > $O/prof_side.log 2>&1
SC_OVERLAP=0 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f -o f -- $B --steps 3 --warmup 1 > $O/pmc_f.log 2>&1
SC_OVERLAP=0 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w -o w -- $B --steps 3 --warmup 1 > $O/pmc_w.log 2>&1
SC_OVERLAP=0 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/pmc_m -o m -- $B --steps 2 --warmup 1 > $O/pmc_m.log 2>&1
cd $R
python tools/pmc_summary.py $(find $O/pmc_f -name "f_counter_collection.csv") $(find $O/pmc_w -name "w_counter_collection.csv") $O/pmc_traffic_summary.json > $O/pmc_traffic.txt 2>&1; head -12 $O/pmc_traffic.txt
python tools/pmc_generic.py $O/pmc_mfma_summary.json "$O/pmc_m/**/m_counter_collection.csv" > $O/pmc_mfma.txt 2>&1; head -12 $O/pmc_mfma.txt
find $O -name "*kernel_trace.csv" -size +20M -delete; find $O -name "*counter_collection.csv" -size +20M -delete
find $O -name "*kernel_stats.csv"; true
