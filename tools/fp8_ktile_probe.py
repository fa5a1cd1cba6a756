#!/usr/bin/env python3
"""Round-5 verdict item 5: the e4m3 NT kernel is byte-for-byte the bf16 schedule with twice the FLOPs per K tile, yet a K tile
takes 2.1 us against 1.45 us.  Fragment path, or clock?  One ViT-L/14 shape (c_fc: 65792 x 4096 x 1024, plain bias epilogue)
on both kernels, on RANDOM operands and on ALL-ZERO operands (no data toggling: the chip holds its clock; MI355X_MICROARCH.md
'DVFS give-back' item 1).  Run under rocprofv3 --pmc (tools/gpu_run.sh pmcx) for GRBM_GUI_ACTIVE (effective clock =
GRBM_GUI_ACTIVE / 8 / wall) and the SQ LDS / MFMA counters; alone it prints wall times.

    python tools/fp8_ktile_probe.py            # DATA=random|zeros|both (default both), N launches each (default 20)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spatial_clip_amd  # noqa: F401,E402
from spatial_clip_amd import ops  # noqa: E402

M, N, K = int(os.environ.get("M", 65792)), int(os.environ.get("NN", 4096)), int(os.environ.get("K", 1024))
n = int(os.environ.get("N", 20))
g = torch.Generator(device="cuda").manual_seed(0)
fl = 2.0 * M * N * K


def timeit(fn, reps=n):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for data in [d for d in ("random", "zeros") if os.environ.get("DATA", "both") in (d, "both")]:
    if data == "random":
        a = torch.randn(M, K, device="cuda", generator=g).bfloat16()
        w = (torch.randn(N, K, device="cuda", generator=g) * 0.03).bfloat16()
    else:
        a = torch.zeros(M, K, device="cuda", dtype=torch.bfloat16)
        w = torch.zeros(N, K, device="cuda", dtype=torch.bfloat16)
    bias = torch.zeros(N, device="cuda")
    a8, sa = ops.quantize_rows_fp8(a)
    w8, sw = ops.quantize_rows_fp8(w)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    tb = timeit(lambda: ops.gemm(ops.NT, ops.EPI_BF16_BIAS, a, w, out, M=M, N=N, K=K, bias=bias))
    t8 = timeit(lambda: ops.gemm_fp8(ops.EPI_BF16_BIAS, a8, sa, w8, sw, out, M=M, N=N, K=K, bias=bias))
    print(f"[{data:6s} operands] {M} x {N} x {K}: bf16 {tb:7.1f} us = {fl / tb / 1e6:6.0f} TFLOP/s;  e4m3 {t8:7.1f} us = {fl / t8 / 1e6:6.0f} TFLOP/s;  "
          f"x{tb / t8:4.2f}", flush=True)
