#!/usr/bin/env python3
"""Per-tile fixed cost vs per-K-tile cost: time(K) at fixed M, N for one epilogue (least-squares line over K)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spatial_clip_amd  # noqa
from spatial_clip_amd import ops
from tools.bench_gemm import run
M = 256 * 197
for name, epi, N in (("bias N=2304", ops.EPI_BF16_BIAS, 2304), ("bf16 N=768", ops.EPI_BF16, 768), ("res N=768", ops.EPI_F32_BIAS_RES, 768),
                     ("gelu N=3072", ops.EPI_GELU_PAIR, 3072)):
    for K in (256, 512, 768, 1536, 3072):
        run(f"{name} K={K}", ops.NT, epi, M, N, K)
