#!/bin/bash
# round 4: SC_OVERLAP=auto -- test, then default bench runs of the three models (what does each stack choose, and the step time)
O=gpurun_out/r4z4; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_model.py -x -q -k "side_stream" > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
show() { python - "$1" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split('/')[-1], d["ms_per_step"], d.get("side_stream"), (d.get("roofline") or {}).get("achieved"))
PY
}
for rep in 1 2; do
  for ov in auto 0 1; do
    if [ $ov = auto ]; then unset SC_OVERLAP; else export SC_OVERLAP=$ov; fi
    timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-loss-delta > $O/head_${ov}_$rep.json 2> $O/head_${ov}_$rep.err || { tail -5 $O/head_${ov}_$rep.err; exit 1; }
    show $O/head_${ov}_$rep.json
  done
done
unset SC_OVERLAP
timeout -k 10 400 python bench.py --model ViT-L-14-genetr --loss spatial --steps 6 --warmup 3 --no-cpu-baseline --no-loss-delta --no-kernel-events > $O/cfg4bf_auto.json 2> $O/cfg4bf_auto.err || { tail -5 $O/cfg4bf_auto.err; exit 1; }
show $O/cfg4bf_auto.json
timeout -k 10 400 python bench.py --model ViT-L-14-genetr --loss spatial --dtype fp8 --steps 6 --warmup 3 --no-cpu-baseline --no-loss-delta --no-kernel-events > $O/cfg4f8_auto.json 2> $O/cfg4f8_auto.err || { tail -5 $O/cfg4f8_auto.err; exit 1; }
show $O/cfg4f8_auto.json
