#!/bin/bash
# round-3 final profile session: bench lines + rocprofv3 kernel stats (single stream / shipped schedule)
O=$PWD/gpurun_out/r3ag; mkdir -p $O
R=$PWD
python bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_line.err; tail -c 400 $O/bench_line.json
python bench.py --loss spatial --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events > $O/bench_line_spatial.json 2> $O/bench_spatial.err
python bench.py --residual-stream bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-events > $O/bench_line_res_stream_bf16.json 2> $O/bench_rs.err
SC_FORCE_DIST=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-loss-delta --no-kernel-events > $O/bench_line_forced_dist_one_rank.json 2> $O/bench_fd.err
cd /tmp && export TMPDIR=/tmp
SC_OVERLAP=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_single -o s -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-events --no-loss-delta > $O/prof_single.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_side -o s -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-events --no-loss-delta > $O/prof_side.log 2>&1
cd $R
find $O -name "*kernel_trace.csv" -size +20M -delete
find $O -name "*kernel_stats.csv" | head
