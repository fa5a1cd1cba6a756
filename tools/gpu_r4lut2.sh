#!/bin/bash
O=gpurun_out/r4lut2; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_fp8.py tests/test_gpu_gemm.py -x -q > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for rep in 1 2 3; do
  for k in 0 1; do
    SC_GELU_LUT=$k SC_OVERLAP=1 timeout -k 10 300 python bench.py --model ViT-L-14-genetr --loss spatial --dtype fp8 --steps 6 --warmup 3 --no-cpu-baseline --no-loss-delta --no-kernel-events > $O/cfg4f8_lut${k}_$rep.json 2> $O/cfg4f8_lut${k}_$rep.err || { tail -5 $O/cfg4f8_lut${k}_$rep.err; exit 1; }
    python - <<PY
import json
d=json.loads(open("$O/cfg4f8_lut${k}_$rep.json").read().strip().splitlines()[-1])
print("configs[4] e4m3 SC_GELU_LUT=$k rep $rep", d["ms_per_step"])
PY
  done
done
