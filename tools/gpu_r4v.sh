#!/bin/bash
# round 4: e4m3 attention branch -- unit tests, then configs[4] fp8 with / without it (interleaved)
O=gpurun_out/r4v; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_fp8.py -x -q -k "section_scales or e4m3_copies or attention_branch or wgrad_fp8_exact or mlp_weight_gradients" > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -3 $O/tests.log
for rep in 1 2; do
  for a in 0 1; do
    SC_FP8_ATTN=$a timeout -k 10 300 python bench.py --model ViT-L-14-genetr --loss spatial --dtype fp8 --steps 8 --warmup 4 --no-cpu-baseline > $O/cfg4_attn${a}_$rep.json 2> $O/cfg4_attn${a}_$rep.err || { tail -5 $O/cfg4_attn${a}_$rep.err; exit 1; }
    python - <<PY
import json
d=json.loads(open("$O/cfg4_attn${a}_$rep.json").read().strip().splitlines()[-1])
print("SC_FP8_ATTN=$a rep $rep", d["ms_per_step"], d.get("loss_delta_vs_fp32_oracle"), d.get("parity"))
PY
  done
done
