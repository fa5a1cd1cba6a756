#!/bin/bash
# round 4 final checks: smoke(), bench line (driver's command), two-rank launcher-free bench rehearsal is in the suite
O=$PWD/gpurun_out/r4t; mkdir -p $O
timeout -k 10 300 python __graft_entry__.py --smoke > $O/smoke.txt 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.txt
timeout -k 10 500 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 1500 $O/bench.json | cut -c1-1500
