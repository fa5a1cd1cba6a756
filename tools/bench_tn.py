#!/usr/bin/env python3
"""TN (weight-gradient layout) main-loop rate in isolation vs the NT layout, and split-K sensitivity on a wgrad shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spatial_clip_amd  # noqa
from spatial_clip_amd import ops
from tools.bench_gemm import run
run("NT 4096^3", ops.NT, ops.EPI_BF16, 4096, 4096, 4096)
run("TN 4096^3 (bf16 out n/a -> f32)", ops.TN, ops.EPI_F32, 4096, 4096, 4096)
run("NT 4096^3 f32 out", ops.NT, ops.EPI_F32, 4096, 4096, 4096)
M = 256 * 197
for sk in (1, 2, 4, 7, 14, 28):
    run(f"c_proj wgrad splitk={sk}", ops.TN, ops.EPI_F32, 768, 3072, M, splitk=sk)
for sk in (7, 14, 28, 32):
    run(f"out_proj wgrad splitk={sk}", ops.TN, ops.EPI_F32, 768, 768, M, splitk=sk)
