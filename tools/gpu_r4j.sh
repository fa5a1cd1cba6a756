#!/bin/bash
# round 4: attention forward, softmax / P.V fused loop (SC_ATTN_FWD_FUSE34=1 build in .ab/fuse34) against the shipped kernel
O=$PWD/gpurun_out/r4j; mkdir -p $O
for rep in 1 2 3; do
  REPS=1 timeout -k 10 200 python tools/bench_attn.py 2>&1 | grep "^fwd" | sed 's/^/shipped  /'
  SC_HIP_LIB=$PWD/.ab/fuse34/libspatialclip_hip.so REPS=1 timeout -k 10 200 python tools/bench_attn.py 2>&1 | grep "^fwd" | sed 's/^/fused34  /'
done | tee $O/fwd_ab.txt
