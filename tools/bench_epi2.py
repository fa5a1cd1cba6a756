#!/usr/bin/env python3
"""Is the epilogue's HBM efficiency limited by scattered 128-B row segments?  Same bytes, different row stride."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spatial_clip_amd  # noqa
from spatial_clip_amd import ops
from tools.bench_gemm import run
M = 256 * 197
for N in (256, 768, 3072):
    run(f"bf16 plain N={N}", ops.NT, ops.EPI_BF16, M * 3072 // N, N, 64)
    run(f"f32 plain N={N}", ops.NT, ops.EPI_F32, M * 3072 // N // 2, N, 64)
