#!/bin/bash
# A/B: alternative build in spatial-clip_amd/lib/alt vs the shipped order
O=$PWD/gpurun_out/r3aa; mkdir -p $O
val() { python -c "
import json,sys
l=[x for x in open(sys.argv[1]) if x.startswith('{')]
d=json.loads(l[-1]); print(sys.argv[2], d['ms_per_step'], d['value'])" $1 "$2"; }
for rep in 1 2 3; do
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-loss-delta --no-kernel-events > $O/fwd_$rep.json 2> $O/fwd_$rep.err; val $O/fwd_$rep.json "shipped build"
  SC_HIP_LIB=$PWD/spatial-clip_amd/lib/alt/libspatialclip_hip.so timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-loss-delta --no-kernel-events > $O/rev_$rep.json 2> $O/rev_$rep.err; val $O/rev_$rep.json "alt build"
done
