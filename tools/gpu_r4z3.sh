#!/bin/bash
# round 4: is the weight-gradient side stream still worth it?  headline (default flags), configs[4] bf16 / e4m3; interleaved
O=gpurun_out/r4z3; mkdir -p $O
run() {  # tag, env overlap, args...
  local tag=$1 ov=$2; shift 2
  SC_OVERLAP=$ov timeout -k 10 400 python bench.py "$@" > $O/${tag}_ov$ov.json 2> $O/${tag}_ov$ov.err || { tail -5 $O/${tag}_ov$ov.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("$O/${tag}_ov$ov.json").read().strip().splitlines()[-1])
print("$tag SC_OVERLAP=$ov", d["ms_per_step"], d.get("roofline", {}).get("achieved"))
PY
}
for rep in 1 2; do
  for ov in 0 1; do run head$rep $ov --steps 20 --warmup 5 --no-cpu-baseline --no-loss-delta; done
done
for rep in 1 2; do
  for ov in 0 1; do run cfg4bf$rep $ov --model ViT-L-14-genetr --loss spatial --steps 6 --warmup 3 --no-cpu-baseline --no-loss-delta --no-kernel-events; done
done
for rep in 1 2; do
  for ov in 0 1; do run cfg4f8$rep $ov --model ViT-L-14-genetr --loss spatial --dtype fp8 --steps 6 --warmup 3 --no-cpu-baseline --no-loss-delta --no-kernel-events; done
done
