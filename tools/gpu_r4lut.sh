#!/bin/bash
# round 4: GELU epilogue by LDS table -- tests, one GEMM alone, then the steps with / without it
O=gpurun_out/r4lut; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_gemm.py -x -q > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
timeout -k 10 200 python - <<'PY'
import os, torch
import spatial_clip_amd
from spatial_clip_amd import ops
g = torch.Generator(device="cuda").manual_seed(0)
for M, N, K in ((50432, 3072, 768), (65792, 4096, 1024)):
    a = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    b = (torch.randn(N, K, device="cuda", generator=g) * 0.05).bfloat16()
    bias = torch.zeros(N, device="cuda")
    o = torch.empty(M, N, device="cuda", dtype=torch.bfloat16); h = torch.empty_like(o)
    for rep in range(2):
        for sw in ("0", "1"):
            os.environ["SC_GELU_LUT"] = sw
            for _ in range(3): ops.gemm(ops.NT, ops.EPI_GELU_GRAD_PAIR, a, b, o, M=M, N=N, K=K, bias=bias, out2=h)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): ops.gemm(ops.NT, ops.EPI_GELU_GRAD_PAIR, a, b, o, M=M, N=N, K=K, bias=bias, out2=h)
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 20 * 1e3
            print(f"c_fc forward M={M} N={N} K={K} SC_GELU_LUT={sw}: {us:7.1f} us  {2.0*M*N*K/us/1e6:6.0f} TFLOP/s", flush=True)
PY
for rep in 1 2 3; do
  for k in 0 1; do
    SC_GELU_LUT=$k SC_OVERLAP=0 timeout -k 10 300 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-loss-delta --no-kernel-events > $O/head_lut${k}_$rep.json 2> $O/head_lut${k}_$rep.err || { tail -5 $O/head_lut${k}_$rep.err; exit 1; }
    python - <<PY
import json
d=json.loads(open("$O/head_lut${k}_$rep.json").read().strip().splitlines()[-1])
print("headline SC_GELU_LUT=$k rep $rep", d["ms_per_step"])
PY
  done
done
for rep in 1 2; do
  for k in 0 1; do
    SC_GELU_LUT=$k SC_OVERLAP=1 timeout -k 10 300 python bench.py --model ViT-L-14-genetr --loss spatial --steps 6 --warmup 3 --no-cpu-baseline --no-loss-delta --no-kernel-events > $O/cfg4_lut${k}_$rep.json 2> $O/cfg4_lut${k}_$rep.err || { tail -5 $O/cfg4_lut${k}_$rep.err; exit 1; }
    python - <<PY
import json
d=json.loads(open("$O/cfg4_lut${k}_$rep.json").read().strip().splitlines()[-1])
print("configs[4] SC_GELU_LUT=$k rep $rep", d["ms_per_step"])
PY
  done
done
