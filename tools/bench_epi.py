#!/usr/bin/env python3
"""Isolate the GEMM epilogue cost: K=64 (one K tile) launches of every epilogue on M=50432 rows."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spatial_clip_amd  # noqa
from spatial_clip_amd import ops
from tools.bench_gemm import run
M = 256 * 197
for K in (64, 768):
    print("K =", K)
    run("bf16 plain N=768", ops.NT, ops.EPI_BF16, M, 768, K)
    run("f32 plain N=768", ops.NT, ops.EPI_F32, M, 768, K)
    run("f32 bias+res N=768", ops.NT, ops.EPI_F32_BIAS_RES, M, 768, K)
    run("bf16 bias N=2304", ops.NT, ops.EPI_BF16_BIAS, M, 2304, K)
    run("gelu pair N=3072", ops.NT, ops.EPI_GELU_PAIR, M, 3072, K)
    run("dgelu N=3072", ops.NT, ops.EPI_BF16_DGELU, M, 3072, K)
    run("bf16 plain N=3072", ops.NT, ops.EPI_BF16, M, 3072, K)
