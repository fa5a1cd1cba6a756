#!/bin/bash
# round 4: bf16 TN kernel with scalar-base DMA addressing (227 instead of 256 VGPRs): exactness, kernel alone, step A/B against the previous build
O=$PWD/gpurun_out/r4s; mkdir -p $O; PREV=$PWD/.ab/prev/libspatialclip_hip.so
timeout -k 10 600 python -m pytest tests/test_gpu_gemm.py -x -q -m gpu -k "tn or wgrad" > $O/tests.txt 2>&1; rc=$?; tail -3 $O/tests.txt
[ $rc -ne 0 ] && exit $rc
for rep in 1 2; do
  SC_HIP_LIB=$PREV timeout -k 10 200 python tools/bench_tn.py 2>&1 | grep -E "TN 4096|c_proj wgrad splitk=7|out_proj wgrad splitk=28" | sed 's/^/prev  /'
  timeout -k 10 200 python tools/bench_tn.py 2>&1 | grep -E "TN 4096|c_proj wgrad splitk=7|out_proj wgrad splitk=28" | sed 's/^/new   /'
done | tee $O/bench_tn.txt
val() { python -c "
import json,sys
l=[x for x in open(sys.argv[1]) if x.startswith('{')]
d=json.loads(l[-1]); print(sys.argv[2], d['ms_per_step'], d['value'])" $1 "$2"; }
for rep in 1 2 3; do
  SC_HIP_LIB=$PREV timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-loss-delta --no-kernel-events > $O/prev_$rep.json 2> $O/prev_$rep.err; val $O/prev_$rep.json "previous TN kernel"
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-loss-delta --no-kernel-events > $O/new_$rep.json 2> $O/new_$rep.err; val $O/new_$rep.json "scalar-base DMA TN kernel"
done
